"""HRNet (backbones/hrnet.py of the reference): BasicBlock :16-56, Bottleneck :59-104, ConvBlock :107-131, TransitionBlockStack :134-154,
DownSampleBlock :157-173, HighResolutionLayer :176-212, HighResolutionFuseStack :215-248, HighResolutionFuseModule :251-309,
HighResolutionModule :312-356, HighResolutionTransitionLayer :359-412, HighResolutionStage :415-460, HighResolutionNet :463-538,
HRNetW48 / HRNetW32 :541-558 -- same classes, attributes and weight names.  Everything is a composition of operators the library already
has (3x3 / 1x1 convolutions, SyncBN with fused ReLU, add + ReLU, concat) plus the aligned-corner bilinear resize
(tf.compat.v1.image.resize(..., align_corners=True), csrc/resize.hip iseg_resize_bilinear_ac_*).

Two behaviours of the reference are kept on purpose:
  * HighResolutionFuseModule.call (:287-309) writes each fused branch back into the list it is still reading from, so branch i > 0 is fused
    from the ALREADY FUSED lower-index branches, not from the module's inputs (the original PyTorch HRNet keeps a separate output list);
  * BasicBlock passes `strides` to both of its convolutions (:23-30); every BasicBlock of this network is built with strides = 1."""
from .. import functional as F
from ..layers.base_layers import Conv2D
from ..layers.normalizations import normalization
from ..nn import Layer
from .resnet_blocks import _bn_relu


def _shape(c):
    return (None, None, None, c)


class BasicBlock(Layer):
    expansion = 1

    def __init__(self, filters, strides=1, downsample=None, name=None):
        super().__init__(name=name)
        self.conv1 = Conv2D(filters, (3, 3), strides=strides, padding="same", use_bias=False, name=f"{name}/conv1")
        self.bn1 = normalization(name=f"{name}/bn1")
        self.conv2 = Conv2D(filters, (3, 3), strides=strides, padding="same", use_bias=False, name=f"{name}/conv2")
        self.bn2 = normalization(name=f"{name}/bn2")
        self.downsample = downsample
        self.strides = strides

    def call(self, inputs, training=None):
        x, residual = F.fork(inputs, 2)
        if self.downsample is not None:
            residual = self.downsample(residual, training=training)
        x = _bn_relu(self.bn1, self.conv1(x), training)
        x = self.bn2(self.conv2(x), training=training)
        return F.add_relu(x, residual)


class Bottleneck(Layer):
    expansion = 4

    def __init__(self, filters, strides=1, downsample=None, name=None):
        super().__init__(name=name)
        self.conv1 = Conv2D(filters, (1, 1), use_bias=False, name=f"{name}/conv1")
        self.bn1 = normalization(name=f"{name}/bn1")
        self.conv2 = Conv2D(filters, (3, 3), strides=strides, padding="same", use_bias=False, name=f"{name}/conv2")
        self.bn2 = normalization(name=f"{name}/bn2")
        self.conv3 = Conv2D(filters * self.expansion, (1, 1), use_bias=False, name=f"{name}/conv3")
        self.bn3 = normalization(name=f"{name}/bn3")
        self.downsample = downsample
        self.strides = strides

    def call(self, inputs, training=None):
        x, residual = F.fork(inputs, 2)
        if self.downsample is not None:
            residual = self.downsample(residual, training=training)
        x = _bn_relu(self.bn1, self.conv1(x), training)
        x = _bn_relu(self.bn2, self.conv2(x), training)
        x = self.bn3(self.conv3(x), training=training)
        return F.add_relu(x, residual)


class ConvBlock(Layer):
    def __init__(self, filters, kernel_size=(1, 1), strides=1, use_relu=True, name=None):
        super().__init__(name=name)
        self.use_relu = use_relu
        self.conv = Conv2D(filters, kernel_size, strides=strides, padding="same", use_bias=False, name=f"{name}/0")
        self.norm = normalization(name=f"{name}/1")

    def call(self, inputs, training=None):
        x = self.conv(inputs)
        if self.use_relu:
            return _bn_relu(self.norm, x, training)
        return self.norm(x, training=training)


class TransitionBlockStack(Layer):
    def __init__(self, filters_list, name=None):
        super().__init__(name=name)
        import torch

        self.blocks = torch.nn.ModuleList([ConvBlock(filters_list[i], (3, 3), strides=2, use_relu=True, name=f"{name}/{i}")
                                           for i in range(len(filters_list))])

    def call(self, inputs, training=None):
        x = inputs
        for block in self.blocks:
            x = block(x, training=training)
        return x


class DownSampleBlock(Layer):
    def __init__(self, filters=None, strides=1, name=None):
        super().__init__(name=name)
        self.conv = Conv2D(filters, (1, 1), strides=strides, use_bias=False, name=f"{name}/0")
        self.norm = normalization(name=f"{name}/1")

    def call(self, inputs, training=None):
        return self.norm(self.conv(inputs), training=training)


class HighResolutionLayer(Layer):
    def __init__(self, block_func, filters, num_blocks, strides=1, name=None):
        super().__init__(name=name)
        self.filters = filters
        self.block_func = block_func
        self.strides = strides
        self.num_blocks = num_blocks
        self.hr_blocks = None

    def build(self, input_shape):
        import torch

        channels = int(input_shape[-1])
        downsample = None
        if self.strides != 1 or channels != self.filters * self.block_func.expansion:
            downsample = DownSampleBlock(filters=self.filters * self.block_func.expansion, strides=self.strides,
                                         name=f"{self.name}/0/downsample")
        blocks = [self.block_func(filters=self.filters, strides=self.strides, downsample=downsample, name=f"{self.name}/0")]
        for i in range(1, self.num_blocks):
            blocks.append(self.block_func(filters=self.filters, name=f"{self.name}/{i}"))
        self.hr_blocks = torch.nn.ModuleList(blocks)
        self.built = True

    def call(self, inputs, training=None):
        x = inputs
        for hr_block in self.hr_blocks:
            x = hr_block(x, training=training)
        return x


class HighResolutionFuseStack(Layer):
    def __init__(self, dest_branch_index=0, src_branch_index=0, channnels_list=[], num_branches=1, name=None):
        super().__init__(name=name)
        import torch

        self.dest_branch_index = dest_branch_index
        self.src_branch_index = src_branch_index
        self.channels_list = list(channnels_list)
        self.num_branches = num_branches
        diff = dest_branch_index - src_branch_index
        layers = []
        for k in range(diff):
            if k == diff - 1:
                layers.append(ConvBlock(self.channels_list[dest_branch_index], (3, 3), use_relu=False, strides=2, name=f"{self.name}/{k}"))
            else:
                layers.append(ConvBlock(self.channels_list[src_branch_index], (3, 3), use_relu=True, strides=2, name=f"{self.name}/{k}"))
        self.fuse_layers = torch.nn.ModuleList(layers)

    def call(self, inputs, training=None):
        x = inputs
        for fuse_layer in self.fuse_layers:
            x = fuse_layer(x, training=training)
        return x


class HighResolutionFuseModule(Layer):
    def __init__(self, multi_scale_output=True, name=None):
        super().__init__(name=name)
        self.multi_scale_output = multi_scale_output
        self.fuse_branches = None

    def build(self, input_shape):
        import torch

        shapes = list(input_shape)
        self.num_branches = len(shapes)
        self.channels_list = [int(s[-1]) for s in shapes]
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:
                    row.append(ConvBlock(self.channels_list[i], use_relu=False, name=f"{self.name}/{i}/{j}"))
                elif j == i:
                    row.append(None)
                else:
                    row.append(HighResolutionFuseStack(i, j, self.channels_list, self.num_branches, name=f"{self.name}/{i}/{j}"))
            rows.append(row)
        self.fuse_branches = rows
        # (None entries cannot live in a ModuleList: register the layers one by one under their reference names)
        for i, row in enumerate(rows):
            for j, layer in enumerate(row):
                if layer is not None:
                    self.add_module(f"fuse_{i}_{j}", layer)
        self.built = True

    def call(self, inputs, training=None):
        x_list = list(inputs)
        nb, n_out = self.num_branches, len(self.fuse_branches)
        sizes = [t.shape[1:3] for t in x_list]
        # Every list slot has several readers (the fuse rows, and the caller for what is returned): one alias per reader, so that the
        # readers' gradients are summed by our own kernel (F.fork).  Slot j is read in its input version by rows i <= j and in its fused
        # version -- written back by row j -- by rows i > j and by the caller.
        slots = []
        for j in range(nb):
            readers = (1 if j == 0 else min(j, n_out - 1) + 1) + (1 if j >= n_out else 0)
            slots.append(list(F.fork(x_list[j], readers)))
        for i in range(n_out):
            y = slots[0].pop() if i == 0 else self.fuse_branches[i][0](slots[0].pop(), training=training)
            for j in range(1, nb):
                x = slots[j].pop()
                if i != j:
                    x = self.fuse_branches[i][j](x, training=training)
                    if j > i:
                        x = F.resize_bilinear(x, sizes[i], align_corners=True)
                y = F.add_relu(y, x) if j == nb - 1 else F.add(y, x)
            # written back into the list the next rows read (the reference's behaviour, see the module docstring)
            assert not slots[i], "fuse bookkeeping: an input alias of this slot was left unread"
            slots[i] = list(F.fork(y, (n_out - 1 - i) + 1))
        out = [s.pop() for s in slots]
        assert not any(slots), "fuse bookkeeping: unread aliases"
        return out


class HighResolutionModule(Layer):
    def __init__(self, block_func, num_block_list, filters_list, multi_scale_output=True, name=None):
        super().__init__(name=name)
        self.block_func = block_func
        self.num_block_list = num_block_list
        self.filters_list = filters_list
        self.multi_scale_output = multi_scale_output
        self.branches = None

    def build(self, input_shape):
        import torch

        shapes = list(input_shape)
        self.num_branhces = len(shapes)
        self.branches = torch.nn.ModuleList([
            HighResolutionLayer(self.block_func, self.filters_list[i], self.num_block_list[i], name=f"{self.name}/branches/{i}")
            for i in range(self.num_branhces)])
        self.fuse_module = HighResolutionFuseModule(self.multi_scale_output, name=f"{self.name}/fuse_layers")
        self.built = True

    def call(self, inputs, training=None):
        x_list = list(inputs)
        for i in range(self.num_branhces):
            x_list[i] = self.branches[i](x_list[i], training=training)
        if self.num_branhces == 1:
            return x_list
        return self.fuse_module(x_list, training=training)


class HighResolutionTransitionLayer(Layer):
    def __init__(self, filters_list=[], name=None):
        super().__init__(name=name)
        self.filters_list = list(filters_list)
        self.transition_layers = None

    def build(self, input_shape):
        shapes = list(input_shape)
        num_in = len(shapes)
        channels_list = [int(s[-1]) for s in shapes]
        layers = []
        for i in range(len(self.filters_list)):
            if i < num_in:
                if self.filters_list[i] != channels_list[i]:
                    layers.append(ConvBlock(filters=self.filters_list[i], kernel_size=(3, 3), use_relu=True, name=f"{self.name}/{i}"))
                else:
                    layers.append(None)
            else:
                sub = [self.filters_list[i] if j == i - num_in else channels_list[-1] for j in range(i + 1 - num_in)]
                layers.append(TransitionBlockStack(sub, name=f"{self.name}/{i}"))
        self.transition_layers = layers
        for i, layer in enumerate(layers):
            if layer is not None:
                self.add_module(f"transition_{i}", layer)
        self.built = True

    def call(self, inputs, training=None):
        x_list = list(inputs)
        num_in = len(x_list)
        # the last input feeds every new branch as well as (possibly) its own transition: one alias per consumer
        uses = [1] * num_in
        for i in range(num_in, len(self.transition_layers)):
            uses[-1] += 1
        if uses[-1] > 1:
            aliases = list(F.fork(x_list[-1], uses[-1]))
            x_list[-1] = aliases.pop()
        y_list = []
        for i, layer in enumerate(self.transition_layers):
            if layer is not None:
                x = aliases.pop() if num_in <= i else x_list[i]
                y_list.append(layer(x, training=training))
            else:
                y_list.append(x_list[i])
        return y_list


class HighResolutionStage(Layer):
    def __init__(self, num_modules, num_block_list, filters_list, block_func=BasicBlock, multi_scale_output=True, name=None):
        super().__init__(name=name)
        filters_list = [filters_list[i] * block_func.expansion for i in range(len(filters_list))]
        self.num_modules = num_modules
        self.block_func = block_func
        self.num_block_list = num_block_list
        self.filters_list = filters_list
        self.multi_scale_output = multi_scale_output
        self.modules_list = None      # (`modules` is a torch.nn.Module method)
        self.transition = HighResolutionTransitionLayer(filters_list, name=f"{self.name}/transition")

    def build(self, input_shape):
        import torch

        mods = []
        for i in range(self.num_modules):
            keep_all = self.multi_scale_output or i < self.num_modules - 1      # multi_scale_output only matters for the last module
            mods.append(HighResolutionModule(block_func=self.block_func, num_block_list=self.num_block_list, filters_list=self.filters_list,
                                             multi_scale_output=keep_all, name=f"{self.name}/{i}"))
        self.modules_list = torch.nn.ModuleList(mods)
        self.built = True

    def call(self, inputs, training=None):
        x = self.transition(inputs, training=training)
        for module in self.modules_list:
            x = module(x, training=training)
        return x


class HighResolutionNet(Layer):
    def __init__(self, stage1_filters=64, stage1_block_func=Bottleneck, stage1_num_blokcs=4, return_endpoints=False, name=None):
        super().__init__(name=name)
        import torch

        self.return_endpoints = return_endpoints
        self.conv1 = Conv2D(64, (3, 3), strides=2, padding="same", use_bias=False, name="conv1")
        self.bn1 = normalization(name="bn1")
        self.conv2 = Conv2D(64, (3, 3), strides=2, padding="same", use_bias=False, name="conv2")
        self.bn2 = normalization(name="bn2")
        self.layer1 = HighResolutionLayer(block_func=stage1_block_func, filters=stage1_filters, num_blocks=stage1_num_blokcs, name="layer1")
        self.stages = torch.nn.ModuleList()

    def add_stage(self, num_modules=1, filters_list=[48, 96], block_func=BasicBlock, num_blocks_list=[4, 4]):
        stage_index = 2 + len(self.stages)
        self.stages.append(HighResolutionStage(num_modules=num_modules, num_block_list=num_blocks_list, filters_list=filters_list,
                                               block_func=block_func, multi_scale_output=True, name=f"stage{stage_index}"))

    def call(self, inputs, training=None):
        x = F.cast_input(inputs)
        x = _bn_relu(self.bn1, self.conv1(x), training)
        x = _bn_relu(self.bn2, self.conv2(x), training)
        x = self.layer1(x, training=training)
        x_list = [x]
        for stage in self.stages:
            x_list = stage(x_list, training=training)
        size = x_list[0].shape[1:3]
        if self.return_endpoints:      # every branch is returned AND resized into the concatenation
            pairs = [F.fork(t, 2) for t in x_list]
            x_list = [p[0] for p in pairs]
            srcs = [p[1] for p in pairs]
        else:
            srcs = x_list
        y_list = [srcs[0]] + [F.resize_bilinear(srcs[i], size, align_corners=True) for i in range(1, len(srcs))]
        y = F.concat(y_list)
        if self.return_endpoints:
            return x_list + [y]
        return y


def HRNetW48(return_endpoints=False):
    net = HighResolutionNet(64, Bottleneck, 4, return_endpoints=return_endpoints)
    net.add_stage(1, [48, 96], BasicBlock, [4, 4])
    net.add_stage(4, [48, 96, 192], BasicBlock, [4, 4, 4])
    net.add_stage(3, [48, 96, 192, 384], BasicBlock, [4, 4, 4, 4])
    return net


def HRNetW32(return_endpoints=False):
    net = HighResolutionNet(64, Bottleneck, 4, return_endpoints=return_endpoints)
    net.add_stage(1, [32, 64], BasicBlock, [4, 4])
    net.add_stage(4, [32, 64, 128], BasicBlock, [4, 4, 4])
    net.add_stage(3, [32, 64, 128, 256], BasicBlock, [4, 4, 4, 4])
    return net
