"""layers/simpledecoder.py of the reference (:8-36): DeepLabV3+-style decoder -- 1x1 (48) on the low-level map, bilinear
resize of the high-level map to it, concat, two 3x3 ConvNormAct."""
from .. import functional as F
from ..nn import Layer
from ..utils.common import resize_image
from .model_builder import ConvNormAct


class SimpleDecoder(Layer):
    def __init__(self, low_level_filters=48, mlp_filters=256, name=None):
        super().__init__(name=name)
        self.low_level_filters = low_level_filters
        self.low_level_entry_conv = ConvNormAct(self.low_level_filters, (1, 1), name=f"{self.name}/low_level_entry_conv")
        self.finetune_conv0 = ConvNormAct(mlp_filters, (3, 3), name=f"{self.name}/finetune_conv0")
        self.finetune_conv1 = ConvNormAct(mlp_filters, (3, 3), name=f"{self.name}/finetune_conv1")

    def call(self, inputs, training=None, **kwargs):
        low_level_features, result_features = tuple(inputs)
        low_level_features = self.low_level_entry_conv(low_level_features, training=training)
        x = resize_image(result_features, size=low_level_features.shape[1:3])
        x = F.cast_to(x, low_level_features.dtype)
        x = F.concat([low_level_features, x])
        x = self.finetune_conv0(x, training=training)
        x = self.finetune_conv1(x, training=training)
        return x
