"""The handful of keras.layers.* classes the reference's hot path instantiates, with the same constructor
keywords, attribute names (.kernel/.bias/.gamma/.beta/.moving_mean/.moving_variance, .strides, .dilation_rate,
.momentum, .epsilon) and Keras weight layouts, backed by the HIP operators of iseg_amd.functional."""
import torch

from .. import functional as F
from .. import kernels as K
from ..nn import Layer


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def get_activation(act):
    """keras.activations.get for the activations the hot path uses; returns (callable|None, fused_code)"""
    if act is None or act is False or act == "linear":
        return None
    if callable(act):
        return act
    if act == "relu":
        return F.relu
    if act == "gelu":
        return F.gelu
    if act in ("swish", "silu"):
        return F.swish
    if act == "sigmoid":
        return F.sigmoid
    raise ValueError(f"activation {act!r} not supported")


def _act_code(fn):
    if fn is None:
        return K.ACT_NONE
    if fn is F.relu:
        return K.ACT_RELU
    if fn is F.gelu:
        return K.ACT_GELU
    return None


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_initializer="glorot_uniform", bias_initializer="zeros",
                 name=None, trainable=True, **kw):
        super().__init__(name=name, trainable=trainable)
        self.units = int(units)
        self.activation = get_activation(activation)
        self.use_bias = use_bias
        self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer
        self.kernel = self.bias = None

    def build(self, input_shape):
        cin = int(input_shape[-1])
        self.kernel = self.add_weight("kernel", (cin, self.units), self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", (self.units,), self.bias_initializer)
        self.built = True

    def call(self, inputs, training=None):
        code = _act_code(self.activation)
        if code is not None:
            return F.dense(inputs, self.kernel, self.bias, code)
        return self.activation(F.dense(inputs, self.kernel, self.bias))


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", dilation_rate=(1, 1), groups=1, activation=None,
                 use_bias=True, kernel_initializer="glorot_uniform", bias_initializer="zeros", name=None, trainable=True, **kw):
        super().__init__(name=name, trainable=trainable)
        self.filters = int(filters)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = padding.lower()
        self.dilation_rate = _pair(dilation_rate)
        self.groups = int(groups)
        self.activation = get_activation(activation)
        self.use_bias = use_bias
        self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer
        self.kernel = self.bias = None

    def build(self, input_shape):
        cin = int(input_shape[-1])
        if cin % self.groups != 0 or self.filters % self.groups != 0:
            raise ValueError(f"Conv2D: {cin} input channels / {self.filters} filters are not divisible by groups={self.groups}")
        self.kernel = self.add_weight("kernel", (*self.kernel_size, cin // self.groups, self.filters), self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", (self.filters,), self.bias_initializer)
        self.built = True

    def call(self, inputs, training=None):
        y = F.conv2d(inputs, self.kernel, self.bias, _pair(self.strides), _pair(self.dilation_rate), self.padding, self.groups)
        return y if self.activation is None else self.activation(y)


class DepthwiseConv2D(Layer):
    def __init__(self, kernel_size, strides=(1, 1), padding="valid", dilation_rate=(1, 1), use_bias=True,
                 depthwise_initializer="glorot_uniform", bias_initializer="zeros", name=None, trainable=True, **kw):
        super().__init__(name=name, trainable=trainable)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = padding.lower()
        self.dilation_rate = _pair(dilation_rate)
        self.use_bias = use_bias
        self.depthwise_initializer, self.bias_initializer = depthwise_initializer, bias_initializer
        self.depthwise_kernel = self.bias = None

    @property
    def kernel(self):
        return self.depthwise_kernel

    def build(self, input_shape):
        c = int(input_shape[-1])
        self.depthwise_kernel = self.add_weight("depthwise_kernel", (*self.kernel_size, c, 1), self.depthwise_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", (c,), self.bias_initializer)
        self.built = True

    def call(self, inputs, training=None):
        st = _pair(self.strides)
        if st[0] != st[1] or self.padding != "same" or self.kernel_size[0] != self.kernel_size[1] or self.kernel_size[0] % 2 == 0:
            raise NotImplementedError("DepthwiseConv2D: odd square kernels, isotropic strides and padding='same' only")
        d = _pair(self.dilation_rate)
        if d[0] != d[1]:
            raise NotImplementedError("DepthwiseConv2D: anisotropic dilation")
        return F.depthwise_conv2d(inputs, self.depthwise_kernel, self.bias, d[0], strides=st[0])


class SeparableConv2D(Layer):
    """keras.layers.SeparableConv2D (depth_multiplier 1): depthwise k x k without bias, then a 1 x 1 convolution with the bias.  Weights in Keras'
    names and layouts: depthwise_kernel [kh, kw, Cin, 1], pointwise_kernel [1, 1, Cin, filters], bias [filters]."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", dilation_rate=(1, 1), activation=None, use_bias=True,
                 depthwise_initializer="glorot_uniform", pointwise_initializer="glorot_uniform", bias_initializer="zeros", name=None,
                 trainable=True, **kw):
        super().__init__(name=name, trainable=trainable)
        self.filters = int(filters)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = padding.lower()
        self.dilation_rate = _pair(dilation_rate)
        self.activation = get_activation(activation)
        self.use_bias = use_bias
        self.depthwise_initializer, self.pointwise_initializer, self.bias_initializer = depthwise_initializer, pointwise_initializer, bias_initializer
        self.depthwise_kernel = self.pointwise_kernel = self.bias = None

    def build(self, input_shape):
        c = int(input_shape[-1])
        self.depthwise_kernel = self.add_weight("depthwise_kernel", (*self.kernel_size, c, 1), self.depthwise_initializer)
        self.pointwise_kernel = self.add_weight("pointwise_kernel", (1, 1, c, self.filters), self.pointwise_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", (self.filters,), self.bias_initializer)
        self.built = True

    def call(self, inputs, training=None):
        st, d = _pair(self.strides), _pair(self.dilation_rate)
        if st[0] != st[1] or d[0] != d[1] or self.padding != "same" or self.kernel_size[0] != self.kernel_size[1] or self.kernel_size[0] % 2 == 0:
            raise NotImplementedError("SeparableConv2D: odd square kernels, isotropic strides / dilation and padding='same' only")
        if self.kernel_size == (1, 1) and st == (1, 1):
            y = F.scale_channels(inputs, self.depthwise_kernel)      # a 1 x 1 depthwise kernel is one factor per channel ([1, 1, C, 1] = C numbers)
        else:
            y = F.depthwise_conv2d(inputs, self.depthwise_kernel, None, d[0], strides=st[0])
        y = F.conv2d(y, self.pointwise_kernel, self.bias, (1, 1), (1, 1), "same", 1)
        return y if self.activation is None else self.activation(y)


class LayerNormalization(Layer):
    def __init__(self, axis=-1, epsilon=1e-3, center=True, scale=True, name=None, trainable=True, **kw):
        super().__init__(name=name, trainable=trainable)
        if axis != -1:
            raise NotImplementedError("LayerNormalization: axis=-1 only")
        self.epsilon = float(epsilon)
        self.gamma = self.beta = None

    def build(self, input_shape):
        c = int(input_shape[-1])
        self.gamma = self.add_weight("gamma", (c,), "ones")
        self.beta = self.add_weight("beta", (c,), "zeros")
        self.built = True

    def call(self, inputs, training=None):
        return F.layer_norm(inputs, self.gamma, self.beta, self.epsilon)


class BatchNormalization(Layer):
    """keras.layers.BatchNormalization(axis=-1, synchronized=...) -- statistics over every axis but the last."""

    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, beta_initializer="zeros",
                 gamma_initializer="ones", moving_mean_initializer="zeros", moving_variance_initializer="ones", synchronized=False,
                 name=None, trainable=True, **kw):
        super().__init__(name=name, trainable=trainable)
        if axis not in (-1, 3):
            raise NotImplementedError("BatchNormalization: channels-last only")
        self.momentum, self.epsilon = float(momentum), float(epsilon)
        self.synchronized = bool(synchronized)
        self.beta_initializer, self.gamma_initializer = beta_initializer, gamma_initializer
        self.moving_mean_initializer, self.moving_variance_initializer = moving_mean_initializer, moving_variance_initializer
        self.gamma = self.beta = self.moving_mean = self.moving_variance = None

    def build(self, input_shape):
        c = int(input_shape[-1])
        self.gamma = self.add_weight("gamma", (c,), self.gamma_initializer)
        self.beta = self.add_weight("beta", (c,), self.beta_initializer)
        self.moving_mean = self.add_state("moving_mean", (c,), self.moving_mean_initializer)
        self.moving_variance = self.add_state("moving_variance", (c,), self.moving_variance_initializer)
        self.built = True

    def call(self, inputs, training=None, fused_relu=False):
        return F.batch_norm(inputs, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.epsilon, self.momentum,
                            bool(training) and self.trainable, relu=fused_relu, sync=self.synchronized)


class Dropout(Layer):
    def __init__(self, rate, name=None, **kw):
        super().__init__(name=name)
        self.rate = float(rate)

    def call(self, inputs, training=None):
        return F.dropout(inputs, self.rate, bool(training))
