"""backbones/eva/{mlp,swiglu,glumlp}.py of the reference: the three feed-forward variants of an EVA block."""
import torch  # noqa: F401

from ... import functional as F
from ...layers.base_layers import Dense, Dropout, LayerNormalization, get_activation
from ...nn import Layer

LAYER_NORM_EPSILON = 1e-6


def _check_hidden(name, hidden):
    """Hidden widths that are not a multiple of 8 (EVA02-large: int(1024 * 8 / 3) = 2730) take the any-width forms of the row kernels -- scalar GLU,
    one-wavefront-per-row LayerNorm (csrc/eva.hip, csrc/norm.hip) -- and the register-staged GEMM (rows of such a tensor are not 16-byte aligned, so
    neither the 16-byte vector kernels nor the LDS-DMA pipeline can take them): correct, not fast.  Only GluMlp's packed halves need the alignment."""
    if name == "GluMlp" and hidden % 8 != 0:
        raise NotImplementedError(f"{name}: half width {hidden} is not a multiple of 8 (the two halves of one Dense output are read in place as 16-byte pieces)")


class Mlp(Layer):
    """mlp.py:12-99: Dense -> activation -> dropout -> [LayerNorm] -> Dense -> dropout"""

    def __init__(self, hidden_filters=None, output_filters=None, activation="gelu", use_bias=True, use_norm=False, dropout_rate=0.0, trainable=True,
                 name=None):
        super().__init__(name=name, trainable=trainable)
        self.hidden_filters, self.output_filters, self.activation = hidden_filters, output_filters, activation
        self.use_bias, self.use_norm, self.dropout_rate = use_bias, use_norm, dropout_rate

    def build(self, input_shape):
        c = int(input_shape[-1])
        hidden = self.hidden_filters or c
        _check_hidden(type(self).__name__, hidden)
        self.fc1 = Dense(hidden, activation=self.activation, use_bias=self.use_bias, name=f"{self.name}/fc1")
        self.drop1 = Dropout(self.dropout_rate, name="drop1")
        self.norm = LayerNormalization(epsilon=LAYER_NORM_EPSILON, name=f"{self.name}/norm") if self.use_norm else None
        self.fc2 = Dense(self.output_filters or c, use_bias=self.use_bias, name=f"{self.name}/fc2")
        self.drop2 = Dropout(self.dropout_rate, name="drop2")
        self.built = True

    def call(self, inputs, training=None):
        x = self.drop1(self.fc1(inputs), training=training)
        if self.norm is not None:
            x = self.norm(x)
        return self.drop2(self.fc2(x), training=training)


class SwiGLU(Layer):
    """swiglu.py:13-100: activation(fc1_g(x)) * fc1_x(x) -> dropout -> [LayerNorm over the hidden units] -> fc2 -> dropout.  The gate's activation is
    whatever the caller hands over: EvaBlock passes its own `activation` ("gelu" by default, block.py:89-96), not swish."""

    def __init__(self, hidden_filters=None, output_filters=None, activation="swish", use_bias=True, use_norm=True, dropout_rate=0.0, trainable=True,
                 name=None):
        super().__init__(name=name, trainable=trainable)
        self.hidden_filters, self.output_filters, self.activation = hidden_filters, output_filters, activation
        self.use_bias, self.use_norm, self.dropout_rate = use_bias, use_norm, dropout_rate
        get_activation(activation)      # (raises for an unknown name, as keras.activations.get would)

    def build(self, input_shape):
        c = int(input_shape[-1])
        hidden = self.hidden_filters or c
        _check_hidden("SwiGLU", hidden)
        self.fc1_g = Dense(hidden, use_bias=self.use_bias, name=f"{self.name}/fc1_g")
        self.fc1_x = Dense(hidden, use_bias=self.use_bias, name=f"{self.name}/fc1_x")
        self.drop1 = Dropout(self.dropout_rate, name="drop1")
        self.norm = LayerNormalization(epsilon=LAYER_NORM_EPSILON, name=f"{self.name}/norm") if self.use_norm else None
        self.fc2 = Dense(self.output_filters or c, use_bias=self.use_bias, name=f"{self.name}/fc2")
        self.drop2 = Dropout(self.dropout_rate, name="drop2")
        self.built = True

    def call(self, inputs, training=None):
        branch_g, branch_x = F.fork(inputs, 2)
        x = F.glu(self.fc1_g(branch_g), self.fc1_x(branch_x), self.activation)
        x = self.drop1(x, training=training)
        if self.norm is not None:
            x = self.norm(x)
        return self.drop2(self.fc2(x), training=training)


class GluMlp(Layer):
    """glumlp.py:12-112: fc1 -> split in two halves -> x1 * activation(x2) (gate_last) -> dropout -> [LayerNorm] -> fc2 -> dropout"""

    def __init__(self, hidden_filters=None, output_filters=None, activation="sigmoid", use_bias=True, use_norm=True, dropout_rate=0.0, use_conv=False,
                 gate_last=True, trainable=True, name=None):
        super().__init__(name=name, trainable=trainable)
        self.hidden_filters, self.output_filters, self.activation = hidden_filters, output_filters, activation
        self.use_bias, self.use_norm, self.dropout_rate, self.gate_last = use_bias, use_norm, dropout_rate, gate_last
        self.use_conv = bool(use_conv)      # (:41-56) fc1 / fc2 as 1 x 1 Conv2D on [N, H, W, C] maps instead of Dense: kernels [1, 1, Cin, Cout]
        get_activation(activation)

    def _fc(self, filters, name):
        if self.use_conv:
            from ...layers.base_layers import Conv2D

            return Conv2D(filters, (1, 1), use_bias=self.use_bias, name=name)
        return Dense(filters, use_bias=self.use_bias, name=name)

    def build(self, input_shape):
        c = int(input_shape[-1])
        hidden = self.hidden_filters or c
        assert hidden % 2 == 0
        _check_hidden("GluMlp", hidden // 2)
        self.fc1 = self._fc(hidden, f"{self.name}/fc1")
        self.drop1 = Dropout(self.dropout_rate, name=f"{self.name}/drop1")
        self.norm = LayerNormalization(epsilon=LAYER_NORM_EPSILON, name=f"{self.name}/norm") if self.use_norm else None
        self.fc2 = self._fc(self.output_filters or c, f"{self.name}/fc2")
        self.drop2 = Dropout(self.dropout_rate, name="drop2")
        self.built = True

    def call(self, inputs, training=None):
        if self.use_conv and inputs.dim() != 4:
            raise ValueError(f"GluMlp(use_conv=True) takes [N, H, W, C] maps (1 x 1 Conv2D projections), got {tuple(inputs.shape)}")
        x = F.glu_packed(self.fc1(inputs), self.activation, gate_last=self.gate_last)
        x = self.drop1(x, training=training)
        if self.norm is not None:
            x = self.norm(x)
        return self.drop2(self.fc2(x), training=training)
