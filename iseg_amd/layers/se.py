"""layers/se.py of the reference: SqueezeAndExcitationModule (:9-47) -- global average -> 1 x 1 conv -> activation -> 1 x 1 conv -> sigmoid -> channel gate."""
from .. import functional as F
from ..nn import Layer
from .base_layers import Conv2D, get_activation
from .nasfpn import _ChannelGateFn, _SigmoidGateFn


class SqueezeAndExcitationModule(Layer):
    def __init__(self, ratio=16, activation="relu", use_bias=True, name=None, trainable=True):
        super().__init__(name=name, trainable=trainable)
        self.ratio, self.use_bias = ratio, use_bias
        self.activation = get_activation(activation)

    def build(self, input_shape):
        filters = int(input_shape[-1])
        self.down_conv = Conv2D(int(filters / self.ratio), (1, 1), use_bias=self.use_bias, name=f"{self.name}/down_conv")
        self.expand_conv = Conv2D(filters, (1, 1), use_bias=self.use_bias, name=f"{self.name}/expand_conv")
        self.built = True

    def call(self, inputs, training=None):
        from .. import nn as _nn

        x, gated = F.fork(inputs, 2)
        g = self.down_conv(F.global_avg_pool(x))
        if self.activation is not None:
            g = self.activation(g)
        g = self.expand_conv(g)
        if _nn.dry_run():
            return gated
        return _ChannelGateFn.apply(gated, _SigmoidGateFn.apply(g))      # x * sigmoid(gate), one factor per (sample, channel)
