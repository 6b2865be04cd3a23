"""backbones/eva/attention.py of the reference (EvaAttention :19-178): multi-head self-attention with rotary position embedding on q and k (the
class token excluded), a fused qkv projection with bias [q_bias | 0 | v_bias] or three separate projections (k without bias), an optional
LayerNorm on the attention output (`use_norm`, EVA's "scale_attention_inner").  Packed [q | k | v] rows all the way: projection -> iseg_qkv_rope in
place (bias + rotation) -> the packed attention operator (online-softmax kernels for head width 64) -> projection.  The reference's
replace_nan_or_inf guards (:128-134,150-160) are identities on finite data and are not reproduced."""
from ... import functional as F
from ...layers.base_layers import Dense, Dropout, LayerNormalization
from ...nn import Layer

LAYER_NORM_EPSILON = 1e-6


class EvaAttention(Layer):
    def __init__(self, num_heads=8, qkv_bias=True, qkv_fused=True, attention_dropout_rate=0.0, projection_dropout_rate=0.0,
                 attention_head_filters=None, use_norm=True, class_token_size=1, trainable=True, name=None):
        super().__init__(name=name, trainable=trainable)
        self.num_heads, self.qkv_bias, self.qkv_fused = int(num_heads), qkv_bias, qkv_fused
        self.attention_dropout_rate, self.projection_dropout_rate = attention_dropout_rate, projection_dropout_rate
        self.attention_head_filters, self.use_norm, self.class_token_size = attention_head_filters, use_norm, int(class_token_size)

    def build(self, input_shape):
        c = int(input_shape[-1])
        head_filters = c // self.num_heads if self.attention_head_filters is None else int(self.attention_head_filters)
        if head_filters * self.num_heads != c:
            # (:57-58 sizes the projections with attention_head_filters * num_heads, but :163 reshapes the attention output to
            # [batch, tokens, CHANNELS]: the reference itself fails for any other product -- an argument error, not a missing feature)
            raise ValueError(f"EvaAttention: attention_head_filters * num_heads = {head_filters * self.num_heads} must equal the input channels {c} "
                             "(the reference reshapes the attention output to the input width, backbones/eva/attention.py:163)")
        self.head_filters = head_filters
        self.attention_scale = head_filters ** -0.5
        if self.qkv_fused:
            self.qkv = Dense(3 * c, use_bias=False, name=f"{self.name}/qkv")
            self.qkv.build((None, None, c))
            self.q_bias = self.add_weight("q_bias", (c,), "zeros")      # (k_bias is a constant zero vector, :76)
            self.v_bias = self.add_weight("v_bias", (c,), "zeros")
        else:
            self.q_proj = Dense(c, use_bias=self.qkv_bias, name=f"{self.name}/q_proj")
            self.k_proj = Dense(c, use_bias=False, name=f"{self.name}/k_proj")
            self.v_proj = Dense(c, use_bias=self.qkv_bias, name=f"{self.name}/v_proj")
            for d in (self.q_proj, self.k_proj, self.v_proj):
                d.build((None, None, c))
        self.attention_dropout = Dropout(self.attention_dropout_rate, name="attention_dropout")
        self.norm = LayerNormalization(epsilon=LAYER_NORM_EPSILON, name=f"{self.name}/norm") if self.use_norm else None
        self.projection_dropout = Dropout(self.projection_dropout_rate, name="projection_dropout")
        self.projection = Dense(c, use_bias=True, name=f"{self.name}/proj")
        self.built = True

    def call(self, inputs, rope=None, training=None):
        c = inputs.shape[-1]
        if self.qkv_fused:
            qkv = F.qkv_rope(self.qkv(inputs), self.q_bias, self.v_bias, rope, self.class_token_size, self.num_heads)
        else:
            qkv = F.dense_group(inputs, [self.q_proj.kernel, self.k_proj.kernel, self.v_proj.kernel],
                                [self.q_proj.bias, None, self.v_proj.bias])
            qkv = F.qkv_rope(qkv, None, None, rope, self.class_token_size, self.num_heads)
        y = F.attention_packed(qkv, self.num_heads, c, c, self.attention_scale, dropout_rate=self.attention_dropout_rate, training=bool(training))
        if self.norm is not None:
            y = self.norm(y)
        y = self.projection(y)
        return self.projection_dropout(y, training=training)
