"""backbones/moat/attention.py of the reference: TrailDense (:104-187) and Attention (:190-339) -- multi-head self-attention over all tokens of a
window with head projections kept as [C, heads, head_size] / [heads, head_size, C] kernels ("weight") and biases, q scaled by head_size^-0.5, softmax
in fp32.  The 2-D relative position embedding (:229-277, reindex_2d_einsum_lookup :54-101) is only reachable with use_pos_emb=True, which
get_backbone never passes by default (feature_extractor.py:47,75); it is not built here and raises."""
from ... import functional as F
from ...nn import Layer

_INIT = ("truncated_normal", 0.02)      # tf.random_normal_initializer(stddev=0.02): parity tests inject weights, training starts from N(0, 0.02)


class Attention(Layer):
    def __init__(self, hidden_size, head_size, relative_position_embedding_type=None, scale_ratio=None, name="attention", trainable=True, **kwargs):
        super().__init__(name=name, trainable=trainable)
        if relative_position_embedding_type is not None:
            raise NotImplementedError("MOAT Attention: relative_position_embedding_type (use_pos_emb=True) is not built; the reference's get_backbone "
                                      "default is use_pos_emb=False (feature_extractor.py:47,75)")
        self.hidden_size, self.head_size = int(hidden_size), int(head_size)
        self.num_heads = self.hidden_size // self.head_size
        self._q_scale = self.head_size ** -0.5

    def build(self, input_shape):
        c, h, d = int(input_shape[-1]), self.num_heads, self.head_size
        self.q_weight = self.add_weight("q/weight", (c, h, d), _INIT)
        self.q_bias = self.add_weight("q/bias", (h, d), "zeros")
        self.k_weight = self.add_weight("k/weight", (c, h, d), _INIT)
        self.k_bias = self.add_weight("k/bias", (h, d), "zeros")
        self.v_weight = self.add_weight("v/weight", (c, h, d), _INIT)
        self.v_bias = self.add_weight("v/bias", (h, d), "zeros")
        self.o_weight = self.add_weight("o/weight", (h, d, self.hidden_size), _INIT)
        self.o_bias = self.add_weight("o/bias", (self.hidden_size,), "zeros")
        self.built = True

    def call(self, query, training=None):
        b, hh, ww, c = query.shape
        h, d = self.num_heads, self.head_size
        x = query.reshape(b, hh * ww, c)
        qkv = F.dense_group(x, [self.q_weight, self.k_weight, self.v_weight], [self.q_bias, self.k_bias, self.v_bias])
        y = F.attention_packed(qkv, h, h * d, h * d, self._q_scale)
        return F.dense(y, self.o_weight, self.o_bias, kshape=(h * d, self.hidden_size))
