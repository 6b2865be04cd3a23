"""layers/common_layers.py of the reference: PatchEmbed (:64-119) -- Conv2D(kernel = stride = patch, padding "SAME") + optional norm."""
from ..nn import Layer
from .base_layers import Conv2D


def to_2d_tuple(v):
    if v is None:
        return None
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class PatchEmbed(Layer):
    def __init__(self, patch_size=(4, 4), weights_patch_size=None, embed_filters=96, norm_layer=None, strides=None, padding="SAME",
                 name=None):
        super().__init__(name=name)
        self.patch_size = to_2d_tuple(patch_size)
        self.weights_patch_size = to_2d_tuple(weights_patch_size)
        self.embed_filters = embed_filters
        self.norm_layer = norm_layer
        self.strides = None if strides is None else to_2d_tuple(strides)
        self.padding = padding

    def build(self, input_shape):
        wps = self.weights_patch_size if self.weights_patch_size is not None else self.patch_size
        if self.strides is None:
            self.strides = wps
        self.proj = Conv2D(self.embed_filters, kernel_size=wps, strides=self.strides, padding=self.padding,
                           name=f"{self.name}/projection")
        self.norm = self.norm_layer(epsilon=1e-5, name=f"{self.name}/norm") if self.norm_layer is not None else None
        self.built = True

    def call(self, inputs, training=None):
        x = self.proj(inputs)
        if self.norm is not None:
            x = self.norm(x)
        return x
