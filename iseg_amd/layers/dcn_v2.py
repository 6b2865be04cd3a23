"""layers/dcn_v2.py of the reference (DCNv2 :16-281): modulated deformable convolution, deformable_groups = 1, stride 1.
offset convolution (kernel_size, "SAME", the layer's dilation) -> 9 x (dy, dx) + 9 mask logits -> modulated bilinear sampling of the zero-padded input
(the C ABI's iseg_dcnv2_sample_fwd / _bwd, csrc/dcnv3.hip) -> one GEMM with the [kh kw C, filters] kernel (+ bias) -> activation.  As in the
reference, `dilation_rate` only dilates the OFFSET convolution; the sampling grid is the undilated 3 x 3 patch (:98-103,140-147)."""
from .. import functional as F
from ..nn import Layer
from .base_layers import _pair, get_activation


class DCNv2(Layer):
    def __init__(self, filters, kernel_size, dilation_rate=1, use_bias=True, kernel_initializer="glorot_uniform", bias_initializer="zeros",
                 kernel_regularizer=None, bias_regularizer=None, use_custom_offset=False, activation=None, use_jit_compile=False, name=None,
                 trainable=True, **kwargs):
        super().__init__(name=name, trainable=trainable)
        if kernel_regularizer is not None or bias_regularizer is not None:
            raise NotImplementedError("DCNv2: weight regularisers are not modelled by this package")
        self.filters, self.kernel_size = int(filters), _pair(kernel_size)
        if self.kernel_size != (3, 3):
            raise NotImplementedError("DCNv2: the sampling kernel is built for 3 x 3 (the only size the reference instantiates, layers/fapn.py:56)")
        self.dilation = _pair(dilation_rate)
        self.use_bias, self.use_custom_offset = use_bias, use_custom_offset
        self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer
        self.activation = get_activation(activation)
        self.use_jit_compile = use_jit_compile      # (XLA switch of the reference: nothing to switch here)

    def build(self, input_shape):
        c = int(input_shape[0][-1] if self.use_custom_offset else input_shape[-1])
        ks = self.kernel_size[0] * self.kernel_size[1]
        self.kernel = self.add_weight("kernel", (*self.kernel_size, c, self.filters), self.kernel_initializer)
        self.bias = self.add_weight("bias", (self.filters,), self.bias_initializer) if self.use_bias else None
        self.offset_kernel = self.add_weight("offset_kernel", (*self.kernel_size, c, 3 * ks), "zeros")
        self.offset_bias = self.add_weight("offset_bias", (3 * ks,), "zeros")
        self.built = True

    def call(self, inputs, training=None):
        if self.use_custom_offset:
            x, offset = tuple(inputs)
        else:
            x, offset = F.fork(inputs, 2)
        c = x.shape[-1]
        offset = F.conv2d(offset, self.offset_kernel, self.offset_bias, (1, 1), self.dilation, "same", 1)
        col = F.dcnv2_sample(x, offset)
        y = F.dense(col, self.kernel, self.bias, kshape=(9 * c, self.filters))
        return y if self.activation is None else self.activation(y)
