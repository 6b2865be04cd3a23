"""backbones/resnet_common.py of the reference: Stack (:24-91), Stack2 (:94-184, stride in the LAST block), ResNet (:187-345,
3x3 deep stem when replace_7x7_conv), constructors (:348-520), build_stacks (:523-560), build_atrous_resnet (:561-588),
apply_multi_grid (:591-598)."""
import torch

from .. import functional as F
from .. import static_strings as ss
from ..layers.base_layers import Conv2D
from ..layers.normalizations import normalization
from ..nn import Layer
from .resnet_blocks import BN_EPSILON, BlockType1, BlockType2, _bn_relu
from .utils.layerwise_decay import decay_layers_lr

DEFAULT_CONV_FUNC = Conv2D


class Stack(Layer):
    def __init__(self, filters, blocks_count, stride1=2, use_bias=True, norm_method=None, custom_block=None,
                 conv_func=DEFAULT_CONV_FUNC, name=None):
        super().__init__(name=name)
        block_func = BlockType1 if custom_block is None else custom_block
        blocks = [block_func(filters, stride=stride1, use_bias=use_bias, norm_method=norm_method, conv_func=conv_func,
                             name=name + "_block1")]
        for i in range(2, blocks_count + 1):
            blocks.append(block_func(filters, conv_shortcut=False, use_bias=use_bias, norm_method=norm_method, conv_func=conv_func,
                                     name=name + "_block" + str(i)))
        self.blocks = torch.nn.ModuleList(blocks)
        self.output_endpoint = stride1 > 1

    @property
    def strides(self):
        return self.blocks[0].strides

    def call(self, inputs, training=None, **kwargs):
        x = inputs
        x_before_stride = x
        x = self.blocks[0](x, training=training)
        for block in list(self.blocks)[1:]:
            x = block(x, training=training)
        if self.output_endpoint:
            return x, x_before_stride
        return x


class Stack2(Layer):
    def __init__(self, filters, blocks_count, stride1=2, use_bias=True, norm_method=None, custom_block=None,
                 conv_func=DEFAULT_CONV_FUNC, name=None):
        super().__init__(name=name)
        block_func = BlockType2 if custom_block is None else custom_block
        blocks = []
        if blocks_count > 1:
            blocks.append(block_func(filters, stride=1, conv_shortcut=True, use_bias=use_bias, norm_method=norm_method,
                                     conv_func=conv_func, name=name + "_block1"))
            for i in range(2, blocks_count):
                blocks.append(block_func(filters, conv_shortcut=False, use_bias=use_bias, norm_method=norm_method,
                                         conv_func=conv_func, name=name + "_block" + str(i)))
            blocks.append(block_func(filters, stride=stride1, conv_shortcut=False, use_bias=use_bias, norm_method=norm_method,
                                     conv_func=conv_func, name=name + "_block" + str(blocks_count)))
        else:
            blocks = [block_func(filters, stride=stride1, conv_shortcut=True, use_bias=use_bias, norm_method=norm_method,
                                 conv_func=conv_func, name=name + "_block1")]
        assert len(blocks) == blocks_count
        self.blocks = torch.nn.ModuleList(blocks)
        self.output_endpoint = stride1 > 1

    @property
    def strides(self):
        return self.blocks[-1].strides

    def call(self, inputs, training=None, **kwargs):
        x = inputs
        for block in list(self.blocks)[:-1]:
            x = block(x, training=training)
        x_before_stride = x
        x = self.blocks[-1](x, training=training)
        if self.output_endpoint:
            return x, x_before_stride
        return x


class ResNet(Layer):
    def __init__(self, stacks, use_bias=True, norm_method=None, conv1_depth_multiplier=1, replace_7x7_conv=False,
                 return_endpoints=False, conv_func=DEFAULT_CONV_FUNC, name="resnet"):
        super().__init__(name=name)
        self.replace_7x7_conv = replace_7x7_conv
        self.conv_func = conv_func
        if self.replace_7x7_conv:
            self.build_3x3_resnet(conv1_depth_multiplier, use_bias, norm_method)
        else:
            self.build_7x7_resnet(conv1_depth_multiplier, use_bias, norm_method)
        # MaxPooling2D(3, strides=2): "same" after the 3x3 stem; ZeroPadding2D(1) + "valid" after the 7x7 stem
        self.pool1_strides = (2, 2)
        self.stacks = torch.nn.ModuleList(stacks)
        self.return_endpoints = return_endpoints

    def build_7x7_resnet(self, depth_multiplier=1, use_bias=True, norm_method=None):
        self.conv1_conv = self.conv_func(int(64 * depth_multiplier), 7, strides=2, use_bias=use_bias, name="conv1_conv")
        self.conv1_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name="conv1_bn")

    def compute_7x7_resnet(self, inputs, training=None):
        # ZeroPadding2D(3) + valid 7x7/s2 (:231-243): the padding is folded into the patch gather
        c = self.conv1_conv
        if not c.built:
            c.build(tuple(inputs.shape))
        x = F.conv2d(inputs, c.kernel, c.bias, tuple(c.strides), tuple(c.dilation_rate), ((3, 3), (3, 3)))
        return _bn_relu(self.conv1_bn, x, training)

    def build_3x3_resnet(self, depth_multiplier=1, use_bias=True, norm_method=None):
        if isinstance(depth_multiplier, tuple):
            depth_multiplier = list(depth_multiplier)
        if not isinstance(depth_multiplier, list):
            depth_multiplier = [depth_multiplier] * 3
        self.conv1_1_conv = self.conv_func(int(64 * depth_multiplier[0]), 3, strides=2, padding="SAME", use_bias=use_bias,
                                           name="conv1_1_conv")
        self.conv1_1_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name="conv1_1_bn")
        self.conv1_2_conv = self.conv_func(int(64 * depth_multiplier[1]), 3, strides=1, padding="SAME", use_bias=use_bias,
                                           name="conv1_2_conv")
        self.conv1_2_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name="conv1_2_bn")
        self.conv1_3_conv = self.conv_func(int(128 * depth_multiplier[2]), 3, strides=1, padding="SAME", use_bias=use_bias,
                                           name="conv1_3_conv")
        self.conv1_3_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name="conv1_3_bn")

    def compute_3x3_resnet(self, inputs, training=None):
        x = _bn_relu(self.conv1_1_bn, self.conv1_1_conv(inputs), training)
        x = _bn_relu(self.conv1_2_bn, self.conv1_2_conv(x), training)
        x = _bn_relu(self.conv1_3_bn, self.conv1_3_conv(x), training)
        return x

    def decay_lr(self, rate=0.99):
        stages = list(self.stacks)
        if self.replace_7x7_conv:
            stems = [self.conv1_1_conv, self.conv1_1_bn, self.conv1_2_conv, self.conv1_2_bn, self.conv1_3_conv, self.conv1_3_bn]
        else:
            stems = [self.conv1_conv, self.conv1_bn]
        stages = stems + stages
        stages.reverse()
        decay_layers_lr(stages, rate=rate)

    def call(self, inputs, training=None, **kwargs):
        endpoints = []
        x = F.cast_input(inputs)
        x = (self.compute_3x3_resnet if self.replace_7x7_conv else self.compute_7x7_resnet)(x, training=training)
        endpoints.append(x)   # OS = 2
        if self.replace_7x7_conv:
            x = F.max_pool2d(x, 3, self.pool1_strides, "same")
        else:
            # ZeroPadding2D(1) + "valid" max pool: x >= 0 after the ReLU, so zero padding and ignored padding agree
            x = F.max_pool2d(x, 3, self.pool1_strides, ((1, 1), (1, 1)))
        for stack in self.stacks:
            x = stack(x, training=training)
            if stack.output_endpoint:
                x, value_before_stride = x
                endpoints.append(value_before_stride)
        endpoints.append(x)
        return endpoints if self.return_endpoints else x


def build_stacks(num_of_blocks=[3, 4, 23, 3], use_bias=True, norm_method=None, slim_behaviour=False, custom_block=None,
                 conv_func=DEFAULT_CONV_FUNC):
    if not slim_behaviour:
        strides = [1, 2, 2, 2]
        stacks_func = Stack
    else:
        strides = [2, 2, 2, 1]
        stacks_func = Stack2
        use_bias = False
    filters_list = [64, 128, 256, 512]
    return [stacks_func(filters=filters_list[i], blocks_count=num_of_blocks[i], stride1=strides[i], use_bias=use_bias,
                        norm_method=norm_method, custom_block=custom_block, conv_func=conv_func, name="conv{}".format(i + 2))
            for i in range(4)]


def get_resnet(resnet_name=ss.RESNET50, num_of_blocks=[3, 4, 6, 3], use_bias=True, norm_method=None, replace_7x7_conv=False,
               slim_behaviour=False, conv1_depth_multiplier=1, custom_block=None, conv_func=DEFAULT_CONV_FUNC,
               return_endpoints=False):
    stacks = build_stacks(num_of_blocks=num_of_blocks, use_bias=use_bias, norm_method=norm_method, slim_behaviour=slim_behaviour,
                          custom_block=custom_block, conv_func=conv_func)
    return ResNet(stacks, use_bias=use_bias, norm_method=norm_method, replace_7x7_conv=replace_7x7_conv,
                  conv1_depth_multiplier=conv1_depth_multiplier, return_endpoints=return_endpoints, conv_func=conv_func,
                  name=resnet_name)


def _ctor(name, blocks):
    def fn(use_bias=True, norm_method=None, replace_7x7_conv=False, slim_behaviour=False, custom_block=None, return_endpoints=False,
           conv_func=DEFAULT_CONV_FUNC):
        return get_resnet(resnet_name=name, num_of_blocks=blocks, use_bias=use_bias, norm_method=norm_method,
                          replace_7x7_conv=replace_7x7_conv, slim_behaviour=slim_behaviour, custom_block=custom_block,
                          return_endpoints=return_endpoints, conv_func=conv_func)

    fn.__name__ = name
    return fn


resnet50 = _ctor(ss.RESNET50, [3, 4, 6, 3])
resnet101 = _ctor(ss.RESNET101, [3, 4, 23, 3])
resnet152 = _ctor(ss.RESNET152, [3, 8, 36, 3])


def build_atrous_resnet(resnet, output_stride=32):
    stacks = resnet.stacks
    if len(stacks) != 4:
        return ValueError("Len of stacks must be 4")
    current_os = 4
    if output_stride == 2:
        resnet.pool1_strides = (1, 1)
    current_atrous_rate = 1
    for stack in stacks:
        for block in stack.blocks:
            if block.strides > 1:
                if current_os >= output_stride:
                    current_atrous_rate *= 2
                    block.strides = 1
                    block.atrous_rates = block.atrous_rates * current_atrous_rate
                else:
                    current_os *= 2
            else:
                block.atrous_rates = block.atrous_rates * current_atrous_rate
    return resnet


def apply_multi_grid(resnet, block_index=3, grids=[1, 2, 4]):
    stack = resnet.stacks[block_index]
    for i in range(len(stack.blocks)):
        stack.blocks[i].atrous_rates = stack.blocks[i].atrous_rates * grids[i]
    return resnet
