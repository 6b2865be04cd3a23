"""layers/aspp.py of the reference (:7-71): concat[image level | 1x1 | 3x3 d=r1 | 3x3 d=r2 | 3x3 d=r3]."""
from .. import functional as F
from ..nn import Layer
from .model_builder import ConvNormAct, ImageLevelBlock


class AtrousSpatialPyramidPooling(Layer):
    def __init__(self, filters=256, dilation_rates=[3, 6, 9], dilation_rates_multiplier=1, use_pixel_level=True,
                 use_image_level=True, name=None):
        super().__init__(name=name)
        self.filters = filters
        self.use_pixel_level = use_pixel_level
        self.use_image_level = use_image_level
        self.dilation_rates = list(dilation_rates)
        self.dilation_rates_multiplier = dilation_rates_multiplier

    def build(self, input_shape):
        if self.use_image_level:
            self.image_level_block = ImageLevelBlock(self.filters, name=f"{self.name}/image_level_block")
        if self.use_pixel_level:
            self.pixel_level_block = ConvNormAct(self.filters, (1, 1), name=f"{self.name}/pixel_level_block")
        convs = []
        for rate in self.dilation_rates:
            rate = rate * self.dilation_rates_multiplier
            convs.append(ConvNormAct(self.filters, (3, 3), dilation_rate=rate, name=f"{self.name}/asp_convs_{rate}"))
        import torch

        self.asp_convs = torch.nn.ModuleList(convs)
        self.built = True

    def call(self, inputs, training=None):
        if self._can_group(training):
            return self._call_grouped(inputs)
        results = []
        branches = int(self.use_image_level) + int(self.use_pixel_level) + len(self.asp_convs)
        xs = list(F.fork(inputs, branches))      # one alias per branch: the branch gradients are summed by our own kernel
        blocks = ([self.image_level_block] if self.use_image_level else []) + ([self.pixel_level_block] if self.use_pixel_level else []) + \
            list(self.asp_convs)
        # independent chains: one HIP stream each inside a captured training step (F.parallel_branches), plain calls otherwise
        results = F.parallel_branches([(lambda b=b, x=x: b(x, training=training)) for b, x in zip(blocks, xs)])
        return F.concat(results)

    # ---- data-parallel training: the branches' SyncBN statistics share one all-reduce (and one in backward) -----------------------
    def _branch_blocks(self):
        blocks = []
        if self.use_image_level:
            blocks.append(self.image_level_block.convbnrelu)
        if self.use_pixel_level:
            blocks.append(self.pixel_level_block)
        return blocks + list(self.asp_convs)

    def _can_group(self, training):
        from .. import dist, nn
        from .base_layers import BatchNormalization

        if not (training and dist.active()) or nn.dry_run():
            return False
        for b in self._branch_blocks():
            if not (isinstance(b.bn, BatchNormalization) and b.bn.synchronized and b.bn.trainable and b.activation is F.relu and
                    b.dropout is None and b.bn.built):
                return False
        return True

    def _call_grouped(self, inputs):
        """conv of every branch first, then ONE statistics exchange for the five BatchNormalizations (F.batch_norm_group)"""
        blocks = self._branch_blocks()
        xs = list(F.fork(inputs, len(blocks)))
        h, w = inputs.shape[1], inputs.shape[2]
        pre = []
        for i, b in enumerate(blocks):
            x = xs[i]
            if self.use_image_level and i == 0:
                x = F.global_avg_pool(x)
            pre.append(b.conv(x))
        outs = list(F.batch_norm_group(pre, [b.bn for b in blocks], relu=True))
        if self.use_image_level:
            outs[0] = F.broadcast_hw(outs[0], h, w)
        return F.concat(outs)
