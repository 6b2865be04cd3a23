"""backbones/resnet_blocks.py of the reference: BlockType1 (:20-108, stride in the first 1x1) and BlockType2 (:111-205, the
"beta"/slim bottleneck: stride in the 3x3, average-pooled identity shortcut when strided).  BN epsilon 1.001e-5."""
from .. import functional as F
from ..layers.base_layers import Conv2D
from ..layers.normalizations import normalization
from ..nn import Layer

BN_EPSILON = 1.001e-5
DEFAULT_CONV_FUNC = Conv2D


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _bn_relu(bn, x, training):
    """BN followed by ReLU in one pass over the activation when the norm layer is a BatchNormalization"""
    if hasattr(bn, "moving_mean"):
        return bn(x, training=training, fused_relu=True)
    return F.relu(bn(x, training=training))


class BlockType1(Layer):
    def __init__(self, filters, kernel_size=3, stride=1, conv_shortcut=True, use_bias=True, norm_method=None,
                 conv_func=DEFAULT_CONV_FUNC, name=None):
        super().__init__(name=name)
        self.conv_func = conv_func
        self.conv_shortcut = conv_shortcut
        if self.conv_shortcut:
            self.shortcut_conv = conv_func(4 * filters, kernel_size=1, strides=stride, use_bias=use_bias, name=name + "_0_conv")
            self.shortcut_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name=name + "_0_bn")
        self.conv1_conv = conv_func(filters, kernel_size=1, strides=stride, use_bias=use_bias, name=name + "_1_conv")
        self.conv1_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name=name + "_1_bn")
        self.conv2_conv = conv_func(filters, kernel_size, padding="SAME", use_bias=use_bias, name=name + "_2_conv")
        self.conv2_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name=name + "_2_bn")
        self.conv3_conv = conv_func(4 * filters, kernel_size=1, use_bias=use_bias, name=name + "_3_conv")
        self.conv3_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name=name + "_3_bn")

    @property
    def strides(self):
        return self.conv1_conv.strides[0]

    @strides.setter
    def strides(self, value):
        value = _pair(value)
        self.conv1_conv.strides = value
        if self.conv_shortcut:
            self.shortcut_conv.strides = value

    @property
    def atrous_rates(self):
        return self.conv2_conv.dilation_rate[0]

    @atrous_rates.setter
    def atrous_rates(self, value):
        if self.conv2_conv.built:
            raise ValueError("conv has been built")
        self.conv2_conv.dilation_rate = _pair(value)

    def call(self, inputs, training=None, **kwargs):
        inputs, skip = F.fork(inputs, 2)      # two consumers: their gradients are summed by our own kernel, not by the engine's add
        if self.conv_shortcut:
            shortcut = self.shortcut_bn(self.shortcut_conv(skip), training=training)
        else:
            shortcut = skip
        x = _bn_relu(self.conv1_bn, self.conv1_conv(inputs), training)
        x = _bn_relu(self.conv2_bn, self.conv2_conv(x), training)
        x = self.conv3_bn(self.conv3_conv(x), training=training)
        return F.add_relu(shortcut, x)


class BlockType2(Layer):
    def __init__(self, filters, kernel_size=3, stride=1, conv_shortcut=True, use_bias=False, norm_method=None,
                 downsample_method="avg", conv_func=DEFAULT_CONV_FUNC, name=None):
        super().__init__(name=name)
        self.conv_shortcut = conv_shortcut
        self.downsample_method = downsample_method
        self.conv_func = conv_func
        if self.conv_shortcut:
            self.shortcut_conv = conv_func(4 * filters, kernel_size=1, strides=stride, use_bias=use_bias, name=name + "_0_conv")
            self.shortcut_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name=name + "_0_bn")
        self.conv1_conv = conv_func(filters, kernel_size=1, use_bias=use_bias, name=name + "_1_conv")
        self.conv1_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name=name + "_1_bn")
        self.conv2_conv = conv_func(filters, kernel_size, strides=stride, padding="SAME", use_bias=use_bias, name=name + "_2_conv")
        self.conv2_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name=name + "_2_bn")
        self.conv3_conv = conv_func(4 * filters, kernel_size=1, use_bias=use_bias, name=name + "_3_conv")
        self.conv3_bn = normalization(epsilon=BN_EPSILON, method=norm_method, name=name + "_3_bn")

    @property
    def strides(self):
        return self.conv2_conv.strides[0]

    @strides.setter
    def strides(self, value):
        value = _pair(value)
        self.conv2_conv.strides = value
        if self.conv_shortcut:
            self.shortcut_conv.strides = value

    @property
    def atrous_rates(self):
        return self.conv2_conv.dilation_rate[0]

    @atrous_rates.setter
    def atrous_rates(self, value):
        self.conv2_conv.dilation_rate = _pair(value)

    def call(self, inputs, training=None, **kwargs):
        inputs, shortcut = F.fork(inputs, 2)      # two consumers: their gradients are summed by our own kernel, not by the engine's add
        if self.conv_shortcut:
            shortcut = self.shortcut_bn(self.shortcut_conv(shortcut), training=training)
        if self.strides > 1:
            st = _pair(self.conv2_conv.strides)
            if "avg" in self.downsample_method:
                shortcut = F.avg_pool2d(shortcut, st, st, "same")
            elif "max" in self.downsample_method:
                shortcut = F.max_pool2d(shortcut, st, st, "same")
            else:
                raise ValueError("Only max or avg are supported")
        x = _bn_relu(self.conv1_bn, self.conv1_conv(inputs), training)
        x = _bn_relu(self.conv2_bn, self.conv2_conv(x), training)
        x = self.conv3_bn(self.conv3_conv(x), training=training)
        return F.add_relu(shortcut, x)
