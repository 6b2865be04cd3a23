"""backbones/intern_image/* of the reference: StemLayer (stem_layer.py:13-68), DownsampleLayer (dowmsample_layer.py:12-42),
MLPLayer (mlp_layer.py:10-58), InternImageLayer (intern_image_layer.py:17-174: pre-norm / post-norm / res-post-norm variants
with layer scale gamma1 / gamma2), InternImageBlock (intern_image_block.py:15-122), InternImage (intern_image.py:15-135) and the
tiny / small variants (:137-165).  The reference defines no "base": intern_image_base (112 channels, depths 4/4/21/4, groups
7/14/28/56, post-norm, layer scale 1.0 -- BASELINE configs[4]) is registered here through the same constructor."""
import numpy as np
import torch

from ... import functional as F
from ...layers.base_layers import Conv2D, Dense, Dropout, LayerNormalization
from ...layers.dcn_v3.dcn_v3 import DeformableConvolutionV3
from ...nn import Layer
from ..backbone_registry import register_backbone

LN_EPS = 1e-6


class StemLayer(Layer):
    def __init__(self, filters=96, activation="gelu", name=None):
        super().__init__(name=name)
        self.filters, self.activation = filters, activation

    def build(self, input_shape):
        self.conv1 = Conv2D(self.filters // 2, kernel_size=3, strides=2, padding="same", name=f"{self.name}/conv1")
        self.norm1 = LayerNormalization(epsilon=LN_EPS, name=f"{self.name}/norm1")
        self.conv2 = Conv2D(self.filters, kernel_size=3, strides=2, padding="same", name=f"{self.name}/conv2")
        self.norm2 = LayerNormalization(epsilon=LN_EPS, name=f"{self.name}/norm2")
        self.built = True

    def call(self, inputs, training=None):
        x = F.gelu(self.norm1(self.conv1(inputs)))
        before_2nd_stride = x
        x = self.norm2(self.conv2(x))
        return x, before_2nd_stride


class DownsampleLayer(Layer):
    def build(self, input_shape):
        c = int(input_shape[-1])
        self.conv = Conv2D(c * 2, kernel_size=3, strides=2, padding="same", use_bias=False, name=f"{self.name}/conv")
        self.norm = LayerNormalization(epsilon=LN_EPS, name=f"{self.name}/norm")
        self.built = True

    def call(self, inputs, training=False):
        return self.norm(self.conv(inputs))


class MLPLayer(Layer):
    def __init__(self, hidden_filters=None, out_filters=None, activation="gelu", dropout_rate=0.0, name=None):
        super().__init__(name=name)
        self.hidden_filters, self.out_filters, self.dropout_rate = hidden_filters, out_filters, dropout_rate

    def build(self, input_shape):
        c = int(input_shape[-1])
        self.fc1 = Dense(int(self.hidden_filters), activation="gelu", name=f"{self.name}/fc1")   # GELU in the GEMM epilogue
        self.fc2 = Dense(self.out_filters or c, name=f"{self.name}/fc2")
        self.dropout = Dropout(self.dropout_rate, name=f"{self.name}/dropout")
        self.built = True

    def call(self, inputs, training=False):
        if self.fc1.built and self.fc2.built and (self.dropout_rate == 0.0 or not training):
            return F.mlp_gelu(inputs, self.fc1.kernel, self.fc1.bias, self.fc2.kernel, self.fc2.bias)       # one tape node
        x = self.dropout(self.fc1(inputs), training=training)
        return self.dropout(self.fc2(x), training=training)


class InternImageLayer(Layer):
    def __init__(self, groups, mlp_ratio=4, dropout_rate=0.0, drop_path_rate=0.0, activation="gelu", use_post_norm=False,
                 layer_scale=False, offset_scale=1.0, depthwise_kernel_size=None, use_res_post_norm=False, center_feature_scale=False,
                 trainable=True, name=None):
        super().__init__(trainable=trainable, name=name)
        self.groups, self.mlp_ratio, self.dropout_rate = groups, mlp_ratio, dropout_rate
        self.drop_path_rate = float(drop_path_rate)
        self.use_post_norm, self.layer_scale, self.offset_scale = use_post_norm, layer_scale, offset_scale
        self.depthwise_kernel_size, self.use_res_post_norm, self.center_feature_scale = depthwise_kernel_size, use_res_post_norm, center_feature_scale
        self.drop_path_masks = None      # parity tests may inject the two per-sample factor vectors

    def build(self, input_shape):
        c = int(input_shape[-1])
        self.norm1 = LayerNormalization(epsilon=LN_EPS, name=f"{self.name}/norm1")
        self.dcn = DeformableConvolutionV3(filters=c, kernel_size=3, depthwise_kernel_size=self.depthwise_kernel_size, strides=1,
                                           padding="same", dilation_rate=1, groups=self.groups, offset_scale=self.offset_scale,
                                           center_feature_scale=self.center_feature_scale, name=f"{self.name}/dcn")
        self.norm2 = LayerNormalization(epsilon=LN_EPS, name=f"{self.name}/norm2")
        self.mlp = MLPLayer(hidden_filters=c * self.mlp_ratio, dropout_rate=self.dropout_rate, name=f"{self.name}/mlp")
        self.gamma1 = self.gamma2 = None
        if self.layer_scale is not None:
            assert not self.use_res_post_norm, "use_res_post_norm and layer_scale can not be used at the same time"
            self.gamma1 = self.add_weight("gamma1", (c,), "ones", trainable=self.trainable)
            self.gamma2 = self.add_weight("gamma2", (c,), "ones", trainable=self.trainable)
        if self.use_res_post_norm:
            self.res_post_norm1 = LayerNormalization(epsilon=LN_EPS, name=f"{self.name}/res_post_norm1")
            self.res_post_norm2 = LayerNormalization(epsilon=LN_EPS, name=f"{self.name}/res_post_norm2")
        self.built = True

    def _post_fusable(self):
        from ... import nn

        ps = [self.norm1.gamma, self.norm1.beta, self.norm2.gamma, self.norm2.beta]
        return (not nn.dry_run() and self.norm1.built and self.norm2.built and all(p is not None and p.requires_grad for p in ps)
                and ps[0].shape[0] % 8 == 0)

    def _scale(self, x, gamma):
        return x if gamma is None else F.scale_channels(x, gamma)

    def call(self, inputs, training=None):
        masks = self.drop_path_masks or (None, None)
        dp = lambda t, i: F.drop_path(t, self.drop_path_rate, bool(training), mask=masks[i])  # noqa: E731
        x, residual = F.fork(inputs, 2)      # residual forks: gradients summed by our own kernel
        if self.use_post_norm and self._post_fusable():
            # LayerNorm + layer scale + drop path + skip connection of each half in one pass forward and one backward (F.layer_norm_post)
            def factors(i):
                if not training or self.drop_path_rate == 0.0:
                    return None
                return masks[i] if masks[i] is not None else F.drop_path_factors(inputs.shape[0], 1.0 - self.drop_path_rate, inputs.device)

            n1, n2 = self.norm1, self.norm2
            x = F.layer_norm_post(self.dcn(x, training=training), n1.gamma, n1.beta, n1.epsilon, self.gamma1, factors(0), residual)
            x, residual = F.fork(x, 2)
            return F.layer_norm_post(self.mlp(x, training=training), n2.gamma, n2.beta, n2.epsilon, self.gamma2, factors(1), residual)
        if self.use_post_norm:
            x = dp(self._scale(self.norm1(self.dcn(x, training=training)), self.gamma1), 0)
            x, residual = F.fork(F.add(residual, x), 2)
            x = dp(self._scale(self.norm2(self.mlp(x, training=training)), self.gamma2), 1)
            return F.add(x, residual)
        if self.use_res_post_norm:
            x = dp(self.res_post_norm1(self.dcn(self.norm1(x), training=training)), 0)
            x, residual = F.fork(F.add(residual, x), 2)
            x = dp(self.res_post_norm2(self.mlp(self.norm2(x), training=training)), 1)
            return F.add(x, residual)
        x = dp(self._scale(self.dcn(self.norm1(x), training=training), self.gamma1), 0)
        x, residual = F.fork(F.add(residual, x), 2)
        x = dp(self._scale(self.mlp(self.norm2(x), training=training), self.gamma2), 1)
        return F.add(x, residual)


class InternImageBlock(Layer):
    def __init__(self, depth, groups, use_downsample=True, mlp_ratio=4, dropout_rate=0.0, drop_path_rate=0.0, activation="gelu",
                 use_post_norm=False, offset_scale=1.0, layer_scale=None, depthwise_kernel_size=None, post_norm_block_ids=None,
                 use_res_post_norm=False, center_feature_scale=False, trainable=True, name=None):
        super().__init__(trainable=trainable, name=name)
        self.depth, self.groups, self.use_downsample = depth, groups, use_downsample
        self.mlp_ratio, self.dropout_rate, self.drop_path_rate = mlp_ratio, dropout_rate, drop_path_rate
        self.use_post_norm, self.offset_scale, self.layer_scale = use_post_norm, offset_scale, layer_scale
        self.depthwise_kernel_size, self.post_norm_block_ids = depthwise_kernel_size, post_norm_block_ids
        self.use_res_post_norm, self.center_feature_scale = use_res_post_norm, center_feature_scale

    def build(self, input_shape):
        self.blocks = torch.nn.ModuleList([
            InternImageLayer(groups=self.groups, mlp_ratio=self.mlp_ratio, dropout_rate=self.dropout_rate,
                             drop_path_rate=self.drop_path_rate[i] if isinstance(self.drop_path_rate, list) else self.drop_path_rate,
                             use_post_norm=self.use_post_norm, layer_scale=self.layer_scale, offset_scale=self.offset_scale,
                             depthwise_kernel_size=self.depthwise_kernel_size, use_res_post_norm=self.use_res_post_norm,
                             center_feature_scale=self.center_feature_scale, name=f"{self.name}/layer/{i}") for i in range(self.depth)])
        self.norm = None
        if not self.use_post_norm or self.center_feature_scale:
            self.norm = LayerNormalization(epsilon=LN_EPS, name=f"{self.name}/norm")
        self.post_norms = None
        if self.post_norm_block_ids is not None:
            self.post_norms = torch.nn.ModuleList([LayerNormalization(epsilon=LN_EPS, name=f"{self.name}/post_norms/{i}")
                                                   for i in range(len(self.post_norm_block_ids))])
        self.downsample = DownsampleLayer(name=f"{self.name}/downsample") if self.use_downsample else None
        self.built = True

    def call(self, inputs, training=None):
        x = inputs
        for i, block in enumerate(self.blocks):
            x = block(x, training=training)
            if self.post_norm_block_ids is not None and (i in self.post_norm_block_ids):
                x = self.post_norms[self.post_norm_block_ids.index(i)](x)
        if self.norm is not None:
            x = self.norm(x)
        x_before_downsample = x
        if self.downsample is not None:
            x = self.downsample(x, training=training)
        return x, x_before_downsample


class InternImage(Layer):
    def __init__(self, stem_filters=64, depths=[3, 4, 18, 5], groups=[3, 6, 12, 24], mlp_ratio=4, dropout_rate=0.0, drop_path_rate=0.2,
                 drop_path_type="linear", activation="gelu", layer_scale=None, offset_scale=1.0, use_post_norm=False,
                 depthwise_kernel_size=None, use_level2_post_norm=False, level2_post_norm_block_ids=None, use_res_post_norm=False,
                 use_center_feature_scale=False, return_endpoints=False, name=None):
        super().__init__(name=name)
        self.stem_filters, self.depths, self.groups, self.mlp_ratio = stem_filters, list(depths), list(groups), int(mlp_ratio)
        self.dropout_rate, self.drop_path_rate, self.drop_path_type = dropout_rate, drop_path_rate, drop_path_type.lower()
        self.layer_scale, self.offset_scale, self.use_post_norm = layer_scale, offset_scale, use_post_norm
        self.depthwise_kernel_size = depthwise_kernel_size
        self.use_level2_post_norm, self.level2_post_norm_block_ids = use_level2_post_norm, level2_post_norm_block_ids
        self.use_res_post_norm, self.use_center_feature_scale = use_res_post_norm, use_center_feature_scale
        self.return_endpoints = return_endpoints

    def build(self, input_shape):
        num_blocks, num_layers = len(self.depths), sum(self.depths)
        self.patch_embed = StemLayer(filters=self.stem_filters, name="patch_embed")
        self.pos_drop = Dropout(self.dropout_rate, name="pos_drop")
        if self.drop_path_type == "linear":
            dpr = [float(x) for x in np.linspace(0.0, self.drop_path_rate, num_layers)]
        else:
            raise ValueError(f"drop_path_type: {self.drop_path_type} not supported")
        blocks = []
        for i in range(num_blocks):
            ids = self.level2_post_norm_block_ids if (self.use_level2_post_norm and i == 2) else None
            blocks.append(InternImageBlock(depth=self.depths[i], groups=self.groups[i], use_downsample=(i < num_blocks - 1),
                                           mlp_ratio=self.mlp_ratio, dropout_rate=self.dropout_rate,
                                           drop_path_rate=dpr[sum(self.depths[:i]):sum(self.depths[:i + 1])],
                                           use_post_norm=self.use_post_norm, layer_scale=self.layer_scale, offset_scale=self.offset_scale,
                                           depthwise_kernel_size=self.depthwise_kernel_size, post_norm_block_ids=ids,
                                           use_res_post_norm=self.use_res_post_norm, center_feature_scale=self.use_center_feature_scale,
                                           name=f"block/{i}"))
        self.blocks = torch.nn.ModuleList(blocks)
        self.built = True

    def call(self, inputs, training=None):
        x = F.cast_input(inputs)
        x, before_2nd_stride_x = self.patch_embed(x, training=training)
        x = self.pos_drop(x, training=training)
        endpoints = [before_2nd_stride_x]
        for blk in self.blocks:
            x, x_before_downsample = blk(x, training=training)
            endpoints.append(x_before_downsample)
        return endpoints if self.return_endpoints else x


def intern_image_tiny(return_endpoints=False):
    return InternImage(stem_filters=64, depths=[4, 4, 18, 4], groups=[4, 8, 16, 32], mlp_ratio=4.0, drop_path_rate=0.2, layer_scale=1.0,
                       offset_scale=1.0, use_post_norm=False, return_endpoints=return_endpoints, name="intern_image_tiny")


def intern_image_small(return_endpoints=False):
    return InternImage(stem_filters=80, depths=[4, 4, 21, 4], groups=[5, 10, 20, 40], mlp_ratio=4.0, drop_path_rate=0.3, layer_scale=1.0,
                       offset_scale=1.0, use_post_norm=True, return_endpoints=return_endpoints, name="intern_image_small")


def intern_image_base(return_endpoints=False):
    """InternImage-B (OpenGVLab: 112 channels, depths 4/4/21/4, groups 7/14/28/56, post-norm, layer scale 1.0, drop path 0.4)"""
    return InternImage(stem_filters=112, depths=[4, 4, 21, 4], groups=[7, 14, 28, 56], mlp_ratio=4.0, drop_path_rate=0.4, layer_scale=1.0,
                       offset_scale=1.0, use_post_norm=True, return_endpoints=return_endpoints, name="intern_image_base")


register_backbone(intern_image_base, "intern_image_base")
