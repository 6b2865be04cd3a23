"""layers/dcn_v3/dcn_v3.py of the reference (:15-150): input projection, depthwise conv -> LN -> GELU branch producing the
sampling offsets and the (soft-maxed) modulation mask, the DCNv3 sampling core (op.py / utils.py -> csrc/dcnv3.hip), optional
centre-feature scale, output projection."""
from ... import functional as F
from ...nn import Layer
from ..base_layers import Dense, DepthwiseConv2D, LayerNormalization

LAYER_NORM_EPSILON = 1e-6


class DeformableConvolutionV3(Layer):
    def __init__(self, filters=64, kernel_size=3, depthwise_kernel_size=None, strides=1, padding="SAME", dilation_rate=1, groups=4,
                 offset_scale=1.0, activation="gelu", center_feature_scale=False, name=None):
        super().__init__(name=name)
        assert filters % groups == 0, "filters must be divisible by groups"
        self.offset_scale, self.filters, self.kernel_size = offset_scale, filters, kernel_size
        self.depthwise_kernel_size = depthwise_kernel_size or kernel_size
        self.strides, self.padding, self.dilation_rate = strides, padding, dilation_rate
        if activation not in ("gelu", None):
            raise NotImplementedError("DeformableConvolutionV3: activation gelu / None")
        self.activation = activation
        self.groups, self.filters_per_group = groups, filters // groups
        self.center_feature_scale = center_feature_scale

    def build(self, input_shape):
        input_channel = int(input_shape[-1])
        k2 = self.kernel_size * self.kernel_size
        self.dw_conv = DepthwiseConv2D(kernel_size=self.depthwise_kernel_size, strides=1, padding=self.padding.lower(),
                                       name=f"{self.name}/dw_conv")
        self.dw_norm = LayerNormalization(epsilon=LAYER_NORM_EPSILON, name=f"{self.name}/dw_conv_norm")
        self.offset = Dense(2 * self.groups * k2, kernel_initializer="zeros", bias_initializer="zeros", name=f"{self.name}/offset")
        self.mask = Dense(self.groups * k2, kernel_initializer="zeros", bias_initializer="zeros", name=f"{self.name}/mask")
        self.input_proj = Dense(input_channel, name=f"{self.name}/input_proj")
        self.output_proj = Dense(self.filters, name=f"{self.name}/output_proj")
        if self.center_feature_scale:      # (:98-102; a plain Dense: this port applies no sigmoid to the scale)
            self.center_feature_scale_proj = Dense(self.groups, name=f"{self.name}/center_feature_scale_proj")
        self.built = True

    def call(self, inputs, training=False):
        xa, xb = F.fork(inputs, 2)      # two consumers each for the input and for x1: gradients summed by our own kernel
        x_proj = self.input_proj(xa)
        x1 = self.dw_norm(self.dw_conv(xb))
        if self.activation == "gelu":
            x1 = F.gelu(x1)
        pad = self.kernel_size // 2 if self.padding.upper() == "SAME" else 0
        ks = (self.kernel_size, self.kernel_size)
        if F.dcnv3_joint_ok(x1, self.offset, self.mask, ks):
            # bf16 storage: the two projections as one product, the sampling kernels reading its column ranges in place (F._DcnJointFn)
            if self.center_feature_scale:
                x1a, x1c = F.fork(x1, 2)
                x_proj, x_proj_skip = F.fork(x_proj, 2)
            else:
                x1a = x1
            x = F.dcnv3_joint(x_proj, x1a, self.offset, self.mask, self.groups, self.filters_per_group, ks, self.strides, self.dilation_rate, pad,
                              self.offset_scale)
        else:
            if self.center_feature_scale:
                x1a, x1b, x1c = F.fork(x1, 3)
                x_proj, x_proj_skip = F.fork(x_proj, 2)
            else:
                x1a, x1b = F.fork(x1, 2)
            offset = self.offset(x1a)
            mask = F.softmax_groups(self.mask(x1b), self.kernel_size * self.kernel_size)
            x = F.dcnv3_core(x_proj, offset, mask, self.groups, self.filters_per_group, ks, self.strides, self.dilation_rate, pad, self.offset_scale)
        if self.center_feature_scale:      # (:138-146) x (1 - s) + x_proj s, one s per (pixel, group)
            x = F.dcn_center_blend(x, x_proj_skip, self.center_feature_scale_proj(x1c), self.groups, self.filters_per_group)
        return self.output_proj(x)
