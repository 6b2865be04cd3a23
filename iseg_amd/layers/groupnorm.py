"""layers/groupnorm.py of the reference (:148-207): moments over (H, W, C/G) per sample and group, then
tf.nn.batch_normalization with per-channel gamma / beta."""
from .. import functional as F
from ..nn import Layer


class GroupNormalization(Layer):
    def __init__(self, groups=32, axis=-1, epsilon=1e-3, center=True, scale=True, beta_initializer="zeros",
                 gamma_initializer="ones", name=None, trainable=True, **kwargs):
        super().__init__(name=name, trainable=trainable)
        if axis not in (-1, 3):
            raise NotImplementedError("GroupNormalization: channels-last only")
        self.groups, self.axis, self.epsilon = int(groups), axis, float(epsilon)
        self.center, self.scale = center, scale
        self.beta_initializer, self.gamma_initializer = beta_initializer, gamma_initializer
        self.gamma = self.beta = None

    def build(self, input_shape):
        dim = int(input_shape[-1])
        if self.groups == -1:
            self.groups = dim
        if dim < self.groups:
            raise ValueError(f"Number of groups ({self.groups}) cannot be more than the number of channels ({dim}).")
        if dim % self.groups != 0:
            raise ValueError(f"Number of groups ({self.groups}) must be a multiple of the number of channels ({dim}).")
        if self.scale:
            self.gamma = self.add_weight("gamma", (dim,), self.gamma_initializer)
        if self.center:
            self.beta = self.add_weight("beta", (dim,), self.beta_initializer)
        self.built = True

    def call(self, inputs, training=None):
        return F.group_norm(inputs, self.gamma, self.beta, self.groups, self.epsilon)
