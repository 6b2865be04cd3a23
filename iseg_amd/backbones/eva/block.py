"""backbones/eva/block.py of the reference (EvaBlock :19-183)."""
from ... import functional as F
from ...layers.base_layers import LayerNormalization
from ...layers.model_builder import get_training_value
from ...nn import Layer
from .attention import EvaAttention
from .mlp import GluMlp, Mlp, SwiGLU

LAYER_NORM_EPSILON = 1e-6


class EvaBlock(Layer):
    def __init__(self, num_heads=8, qkv_bias=True, qkv_fused=True, mlp_ratio=4.0, swiglu_mlp=False, scale_mlp=False, scale_attention_inner=False,
                 attention_dropout_rate=0.0, projection_dropout_rate=0.0, drop_path_rate=0.0, init_values=None, activation="gelu",
                 attention_head_filters=None, use_post_norm=False, class_token_size=1, trainable=True, name=None):
        super().__init__(name=name, trainable=trainable)
        self.num_heads, self.qkv_bias, self.qkv_fused, self.mlp_ratio = num_heads, qkv_bias, qkv_fused, mlp_ratio
        self.swiglu_mlp, self.scale_mlp, self.scale_attention_inner = swiglu_mlp, scale_mlp, scale_attention_inner
        self.attention_dropout_rate, self.projection_dropout_rate = attention_dropout_rate, projection_dropout_rate
        self.drop_path_rate, self.init_values, self.activation = float(drop_path_rate), init_values, activation
        self.attention_head_filters, self.use_post_norm, self.class_token_size = attention_head_filters, use_post_norm, class_token_size
        self.drop_path_mask = None      # parity tests may inject the per-sample factor vector

    def build(self, input_shape):
        c = int(input_shape[-1])
        self.norm1 = LayerNormalization(epsilon=LAYER_NORM_EPSILON, name=f"{self.name}/norm1")
        self.attention = EvaAttention(num_heads=self.num_heads, qkv_bias=self.qkv_bias, qkv_fused=self.qkv_fused,
                                      attention_dropout_rate=self.attention_dropout_rate, projection_dropout_rate=self.projection_dropout_rate,
                                      attention_head_filters=self.attention_head_filters, use_norm=self.scale_attention_inner,
                                      class_token_size=self.class_token_size, name=f"{self.name}/attn")
        self.norm2 = LayerNormalization(epsilon=LAYER_NORM_EPSILON, name=f"{self.name}/norm2")
        hidden = int(c * self.mlp_ratio)
        mlp_name = f"{self.name}/mlp"
        if self.swiglu_mlp:
            if self.scale_mlp:      # (:89-96: the gate's activation is the block's `activation`, "gelu" unless the caller says otherwise)
                self.mlp = SwiGLU(hidden_filters=hidden, use_norm=True, activation=self.activation, dropout_rate=self.projection_dropout_rate,
                                  name=mlp_name)
            else:
                self.mlp = GluMlp(hidden_filters=hidden * 2, use_norm=False, activation="swish", dropout_rate=self.projection_dropout_rate,
                                  name=mlp_name)
        else:
            self.mlp = Mlp(hidden_filters=hidden, activation=self.activation, use_norm=self.scale_mlp, dropout_rate=self.projection_dropout_rate,
                           name=mlp_name)
        self.gamma_1 = self.gamma_2 = None
        if self.init_values is not None:
            self.gamma_1 = self.add_weight("gamma_1", (c,), float(self.init_values))
            self.gamma_2 = self.add_weight("gamma_2", (c,), float(self.init_values))
        self.built = True

    def call(self, inputs, rope=None, training=None):
        training = get_training_value(training)
        x, residual = F.fork(inputs, 2)
        if not self.use_post_norm:
            x = self.norm1(x)
        x = self.attention(x, rope=rope, training=training)
        if self.use_post_norm:
            x = self.norm1(x)
        if self.gamma_1 is not None:
            x = F.scale_channels(x, self.gamma_1)
        x = F.drop_path(x, self.drop_path_rate, training, mask=self.drop_path_mask)
        x, residual = F.fork(F.add(x, residual), 2)
        if not self.use_post_norm:
            x = self.norm2(x)
        x = self.mlp(x, training=training)
        if self.use_post_norm:
            x = self.norm2(x)
        if self.gamma_2 is not None:
            x = F.scale_channels(x, self.gamma_2)
        return F.add(x, residual)      # (:180-181: no drop path on the second branch in the reference)
