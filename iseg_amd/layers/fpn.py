"""layers/fpn.py of the reference (:16-61): top-down pathway.  The coarsest map is used raw; every finer level is
`ConvNormAct(1x1)(replace_nan_or_inf(f_i))` plus the bilinearly up-sampled running map.  Returns all levels, fine -> coarse."""
import torch

from .. import functional as F
from ..nn import Layer
from ..utils.common import resize_image
from .model_builder import ConvNormAct


class FeaturePyramidNetwork(Layer):
    def __init__(self, skip_conv_filters=256, trainable=True, name=None):
        super().__init__(name=name, trainable=trainable)
        self.skip_conv_filters = skip_conv_filters

    def build(self, input_shape):
        feature_map_shapes = input_shape
        convs = [ConvNormAct(self.skip_conv_filters, name=f"{self.name}/skip_conv_filters{i}")
                 for i in range(len(feature_map_shapes) - 1)]
        self.skip_convs = torch.nn.ModuleList(convs)
        self.built = True

    @staticmethod
    def _level_fusable(block, x, training):
        """training-mode BatchNorm + ReLU with nothing in between (no dropout), channel count the 16-byte kernels take"""
        from .. import nn
        from .base_layers import BatchNormalization

        bn = block.bn
        return (not nn.dry_run() and bool(training) and isinstance(bn, BatchNormalization) and bn.built and bn.trainable
                and block.activation is F.relu and block.dropout is None and block.conv.filters % 8 == 0 and x.shape[-1] == block.conv.filters)

    def call(self, inputs, training=None):
        feature_map_list = list(inputs)
        x = feature_map_list[-1]
        result_endpoints = [x]
        for i in range(len(self.skip_convs) - 1, -1, -1):
            skip_feature = F.replace_nan_or_inf(feature_map_list[i], 0.0)
            block = self.skip_convs[i]
            if self._level_fusable(block, x, training):
                # BatchNorm + ReLU of the lateral block, the up-sampling of the running map and the sum in one pass each way
                z = block.conv(skip_feature)
                bn = block.bn
                x = F.batch_norm_relu_upsample_add(z, x, bn.gamma, bn.beta, bn.moving_mean, bn.moving_variance, bn.epsilon, bn.momentum,
                                                   sync=bn.synchronized)
            else:
                skip_feature = block(skip_feature, training=training)
                x = resize_image(x, size=skip_feature.shape[1:3])
                x = F.add(x, skip_feature)
            if i > 0:      # this level is returned AND feeds the next one: fork (gradients summed by our own kernel)
                out, x = F.fork(x, 2)
                result_endpoints.append(out)
            else:
                result_endpoints.append(x)
        result_endpoints.reverse()
        return result_endpoints
