"""layers/rmsnorm.py of the reference (:8-29): x * rsqrt(mean(x^2) + eps) * (1 + scale), statistics in float32."""
from .. import functional as F
from ..nn import Layer


class RMSNormalization(Layer):
    def __init__(self, epsilon=1e-6, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.epsilon = float(epsilon)
        self.scale = None

    def build(self, input_shape):
        self.scale = self.add_weight("scale", (int(input_shape[-1]),), "zeros")
        self.built = True

    def call(self, x, training=None):
        return F.rms_norm(x, self.scale, self.epsilon)
