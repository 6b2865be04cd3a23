"""layers/fapn.py of the reference: FeatureSelectionModule (:13-41), FeatureAlignment (:44-80), FeatureAlignedPyramidNet (:83-140) -- the FaPN
decoder: every skip feature is re-weighted by a squeeze-and-excitation gate and projected, the coarser pyramid level is up-sampled bilinearly and
ALIGNED to it by a DCNv2 whose offsets come from both, then added."""
import torch

from .. import functional as F
from ..nn import Layer
from .base_layers import Conv2D, Dense
from .dcn_v2 import DCNv2
from .se import SqueezeAndExcitationModule


class FeatureSelectionModule(SqueezeAndExcitationModule):
    def __init__(self, filters=128, activation="relu", name=None, trainable=True):
        super().__init__(ratio=1, activation=activation, use_bias=False, name=name, trainable=trainable)
        self.filters = filters

    def build(self, input_shape):
        self.conv = Conv2D(self.filters, (1, 1), use_bias=False, name=f"{self.name}/conv")
        super().build(input_shape)

    def call(self, inputs, training=None):
        a, b = F.fork(inputs, 2)
        x = F.add(super().call(a, training=training), b)
        return self.conv(x)


class FeatureAlignment(Layer):
    def __init__(self, filters=128, name=None, trainable=True):
        super().__init__(name=name, trainable=trainable)
        self.filters = filters

    def build(self, input_shape):
        self.lateral_conv = FeatureSelectionModule(filters=self.filters, name=f"{self.name}/lateral_conv", trainable=self.trainable)
        self.offset_conv = Conv2D(self.filters, (1, 1), use_bias=False, name=f"{self.name}/offset_conv", trainable=self.trainable)
        self.depack_l2 = DCNv2(self.filters, (3, 3), use_custom_offset=True, use_jit_compile=True, name=f"{self.name}/depack_l2", trainable=self.trainable)
        self.built = True

    def call(self, inputs, training=None):
        feats_large, feats_small = inputs
        feats_up = F.resize_bilinear(feats_small, (feats_large.shape[1], feats_large.shape[2]))
        arm_a, arm_b = F.fork(self.lateral_conv(feats_large, training=training), 2)
        up_a, up_b, up_c = F.fork(feats_up, 3)
        offset = self.offset_conv(F.concat([arm_a, F.add(up_a, up_b)]))      # tf.concat([feats_arm, feats_up * 2], axis=-1)
        feat_align = F.relu(self.depack_l2([up_c, offset], training=training))
        return F.add(feat_align, arm_b)


class FeatureAlignedPyramidNet(Layer):
    def __init__(self, skip_conv_filters=256, trainable=True, warp_coarse_feature=False, name=None):
        super().__init__(name=name, trainable=trainable)
        self.skip_conv_filters = skip_conv_filters
        self.warp_coarse_feature = warp_coarse_feature

    def build(self, input_shape):
        self.align_modules = torch.nn.ModuleList([FeatureAlignment(self.skip_conv_filters, name=f"{self.name}/skip_conv_filters{i}", trainable=self.trainable)
                                                  for i in range(len(input_shape) - 1)])
        self.coarse_warp_conv = Dense(self.skip_conv_filters, name=f"{self.name}/coarse_warp_conv") if self.warp_coarse_feature else None
        self.built = True

    def call(self, inputs, training=None):
        feature_map_list = list(inputs)
        x = feature_map_list[-1]
        if self.coarse_warp_conv is not None:
            x = self.coarse_warp_conv(x)
        result_endpoints = []
        for i in range(len(self.align_modules) - 1, -1, -1):
            x, keep = F.fork(x, 2)
            result_endpoints.append(keep)
            x = self.align_modules[i]([feature_map_list[i], x], training=training)
        result_endpoints.append(x)
        result_endpoints.reverse()
        return result_endpoints
