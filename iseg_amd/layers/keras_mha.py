"""keras.layers.MultiHeadAttention as backbones/vit.py:142-147,166 of the reference uses it (self-attention, no mask):
query / key / value kernels [C, heads, key_dim] + bias [heads, key_dim]; q * key_dim^-0.5; softmax over keys; dropout on the
probabilities; output kernel [heads, key_dim, C] + bias [C].  Weight names follow Keras (query/key/value/attention_output)."""
from .. import functional as F
from ..nn import Layer


class MultiHeadAttention(Layer):
    def __init__(self, num_heads, key_dim, value_dim=None, dropout=0.0, use_bias=True, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", name=None, trainable=True, **kwargs):
        super().__init__(name=name, trainable=trainable)
        self.num_heads, self.key_dim = int(num_heads), int(key_dim)
        self.value_dim = int(value_dim) if value_dim else int(key_dim)
        self.dropout, self.use_bias = float(dropout), use_bias
        self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer

    def build(self, input_shape):
        c = int(input_shape[-1])
        h, dk, dv = self.num_heads, self.key_dim, self.value_dim
        self.query_kernel = self.add_weight("query/kernel", (c, h, dk), self.kernel_initializer)
        self.key_kernel = self.add_weight("key/kernel", (c, h, dk), self.kernel_initializer)
        self.value_kernel = self.add_weight("value/kernel", (c, h, dv), self.kernel_initializer)
        self.output_kernel = self.add_weight("attention_output/kernel", (h, dv, c), self.kernel_initializer)
        self.query_bias = self.key_bias = self.value_bias = self.output_bias = None
        if self.use_bias:
            self.query_bias = self.add_weight("query/bias", (h, dk), self.bias_initializer)
            self.key_bias = self.add_weight("key/bias", (h, dk), self.bias_initializer)
            self.value_bias = self.add_weight("value/bias", (h, dv), self.bias_initializer)
            self.output_bias = self.add_weight("attention_output/bias", (c,), self.bias_initializer)
        self.built = True

    def call(self, query, value=None, key=None, training=None):
        if (value is not None and value is not query) or (key is not None and key is not query):
            raise NotImplementedError("MultiHeadAttention: self-attention only (query is value is key)")
        b, t, c = query.shape
        h, dk, dv = self.num_heads, self.key_dim, self.value_dim
        # the three projections as one node: column blocks of qkv written / read in place, data gradients accumulated in the GEMM epilogue
        qkv = F.dense_group(query, [self.query_kernel, self.key_kernel, self.value_kernel], [self.query_bias, self.key_bias, self.value_bias])
        x = F.attention_packed(qkv, h, h * dk, h * dv, float(dk) ** -0.5, dropout_rate=self.dropout, training=bool(training))
        return F.dense(x, self.output_kernel, self.output_bias, kshape=(h * dv, c))
