"""Canonical model compositions: the reference defines heads (layers/aspp.py, layers/fpn.py,
layers/simpledecoder.py) but composes them only in downstream user projects (SegManaged.head is set by the user,
layers/core_model_ext.py:91).  These are the <= 15-line compositions the BASELINE configs name."""
from .layers.aspp import AtrousSpatialPyramidPooling
from .layers.core_model_ext import SegManaged
from .layers.model_builder import ConvNormAct
from .nn import Layer


class ASPPHead(Layer):
    """endpoints[-1] -> ASPP(256, [3,6,9] x 32/output_stride) -> ConvNormAct(256, 1x1, dropout 0.1)  (the CommonEndBlock.end_conv
    pattern, layers/model_builder.py:285-288); SegManaged adds logits_conv + bilinear upsample + float32."""

    def __init__(self, filters=256, output_stride=32, dropout_rate=0.1, name="aspp_head"):
        super().__init__(name=name)
        self.aspp = AtrousSpatialPyramidPooling(filters, dilation_rates=[3, 6, 9], dilation_rates_multiplier=max(32 // output_stride, 1),
                                                name=f"{self.name}/aspp")
        self.end_conv = ConvNormAct(filters, (1, 1), dropout_rate=dropout_rate, name=f"{self.name}/end_conv")

    def call(self, inputs, training=None):
        x = inputs[-1] if isinstance(inputs, (list, tuple)) else inputs
        x = self.aspp(x, training=training)
        return self.end_conv(x, training=training)


def convnext_tiny_aspp(num_class=21, output_stride=32, build_input_size=(512, 512), drop_path_rate=None, dropout_rate=0.1,
                       layer_scale_init_value=None):
    """BASELINE config 2: ConvNeXt-T + ASPP."""
    from .backbones import convnext as cx

    custom = None
    if drop_path_rate is not None or layer_scale_init_value is not None:
        def custom(return_endpoints=False):
            return cx.ConvNeXt(depths=[3, 3, 9, 3], filters_list=[96, 192, 384, 768], return_endpoints=return_endpoints,
                               drop_path_rate=0.1 if drop_path_rate is None else drop_path_rate,
                               layer_scale_init_value=1e-6 if layer_scale_init_value is None else layer_scale_init_value)
    model = SegManaged(backbone_name="convnext_tiny", backbone_custom_fn=custom, output_stride=output_stride, num_class=num_class,
                       build_input_size=build_input_size, name="seg")
    model.head = ASPPHead(256, output_stride=output_stride, dropout_rate=dropout_rate)
    model.build_with_dummy()
    return model
