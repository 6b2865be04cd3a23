"""Canonical model compositions: the reference defines heads (layers/aspp.py, layers/fpn.py,
layers/simpledecoder.py) but composes them only in downstream user projects (SegManaged.head is set by the user,
layers/core_model_ext.py:91).  These are the <= 15-line compositions the BASELINE configs name."""
from .layers.aspp import AtrousSpatialPyramidPooling
from .layers.core_model_ext import SegManaged
from . import functional as F
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


class FPNHead(Layer):
    """BASELINE config 3: FeaturePyramidNetwork(skip_conv_filters = C_top)(endpoints[1:]) -> finest level (OS4)
    -> ConvNormAct(256, 1x1)."""

    def __init__(self, top_filters, filters=256, name="fpn_head"):
        super().__init__(name=name)
        from .layers.fpn import FeaturePyramidNetwork

        self.fpn = FeaturePyramidNetwork(skip_conv_filters=top_filters, name=f"{self.name}/fpn")
        self.end_conv = ConvNormAct(filters, (1, 1), name=f"{self.name}/end_conv")

    def call(self, inputs, training=None):
        levels = self.fpn(list(inputs)[1:], training=training)
        return self.end_conv(levels[0], training=training)


class FaPNHead(Layer):
    """FeatureAlignedPyramidNet (layers/fapn.py:83-140) in the place of FPN: the coarsest endpoint must already carry `top_filters` channels
    (warp_coarse_feature=False), every finer level is aligned to it through DCNv2 -> finest level -> ConvNormAct(256, 1x1)."""

    def __init__(self, top_filters, filters=256, name="fapn_head"):
        super().__init__(name=name)
        from .layers.fapn import FeatureAlignedPyramidNet

        self.fapn = FeatureAlignedPyramidNet(skip_conv_filters=top_filters, name=f"{self.name}/fapn")
        self.end_conv = ConvNormAct(filters, (1, 1), name=f"{self.name}/end_conv")

    def call(self, inputs, training=None):
        levels = self.fapn(list(inputs)[1:], training=training)
        return self.end_conv(levels[0], training=training)


class SimpleDecoderHead(Layer):
    """BASELINE config 4: ViT returns one endpoint; SimpleDecoder(48, 256)((e, ConvNormAct(256, 1x1)(e)))."""

    def __init__(self, filters=256, low_level_filters=48, name="decoder_head"):
        super().__init__(name=name)
        from .layers.simpledecoder import SimpleDecoder

        self.high_conv = ConvNormAct(filters, (1, 1), name=f"{self.name}/high_conv")
        self.decoder = SimpleDecoder(low_level_filters, filters, name=f"{self.name}/decoder")

    def call(self, inputs, training=None):
        e = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        low, high = F.fork(e, 2)      # the endpoint feeds both decoder inputs
        return self.decoder((low, self.high_conv(high, training=training)), training=training)


class LastEndpointDecoderHead(SimpleDecoderHead):
    """SimpleDecoderHead on the LAST endpoint of a plain-ViT style backbone that returns one endpoint per block (EVA: [class token, patch embedding,
    block 0, ...], backbones/eva/eva.py:297-310)"""

    def call(self, inputs, training=None):
        return super().call([list(inputs)[-1]], training=training)


def _managed(backbone_name, head, num_class, output_stride, build_input_size, backbone_custom_fn=None):
    model = SegManaged(backbone_name=backbone_name, backbone_custom_fn=backbone_custom_fn, output_stride=output_stride,
                       num_class=num_class, build_input_size=build_input_size, name="seg")
    model.head = head
    model.build_with_dummy()
    return model


def resnet50_aspp(num_class=21, output_stride=32, build_input_size=(256, 256), dropout_rate=0.1):
    """BASELINE config 1: ResNet-50 (slim/beta) + ASPP"""
    return _managed("resnet50", ASPPHead(256, output_stride=output_stride, dropout_rate=dropout_rate), num_class, output_stride,
                    build_input_size)


def swin_tiny_fapn(num_class=21, build_input_size=(512, 512)):
    """Swin-T + the FaPN decoder (layers/fapn.py)"""
    return _managed("swin_tiny_224", FaPNHead(top_filters=768), num_class, 32, build_input_size)


def eva02_tiny_simple_decoder(num_class=21, build_input_size=(448, 448)):
    """EVA02-tiny (backbones/eva/eva.py:441-467, patch 14) + SimpleDecoder on its last endpoint"""
    return _managed("eva02_tiny", LastEndpointDecoderHead(), num_class, 14, build_input_size)


def swin_tiny_fpn(num_class=21, build_input_size=(512, 512)):
    """BASELINE config 3: Swin-T + FPN"""
    return _managed("swin_tiny_224", FPNHead(top_filters=768), num_class, 32, build_input_size)


def vit_base_simple_decoder(num_class=21, build_input_size=(512, 512)):
    """BASELINE config 4: ViT-B/16 + SimpleDecoder (evaluated with sliding-window inference)"""
    return _managed("vit_base", SimpleDecoderHead(), num_class, 16, build_input_size)


def intern_image_base_aspp(num_class=21, build_input_size=(512, 512), dropout_rate=0.1):
    """BASELINE config 5: InternImage-B (DCNv3) + ASPP"""
    from .backbones import intern_image  # noqa: F401  (registers intern_image_base)

    return _managed("intern_image_base", ASPPHead(256, output_stride=32, dropout_rate=dropout_rate), num_class, 32, build_input_size)


def convnext_v2_tiny_aspp(num_class=21, output_stride=32, build_input_size=(512, 512), dropout_rate=0.1):
    """the BASELINE config-2 composition with the ConvNeXt V2 backbone (backbones/convnext_v2.py: GRN instead of layer scale)"""
    return _managed("convnext_v2_tiny", ASPPHead(256, output_stride=output_stride, dropout_rate=dropout_rate), num_class, output_stride,
                    build_input_size)


def hrnet_w32_aspp(num_class=21, build_input_size=(512, 512), dropout_rate=0.1):
    """HRNet-W32 (backbones/hrnet.py: the concatenated map at stride 4 is the last endpoint) + the ASPP head of the BASELINE composition"""
    return _managed("hrnet_w32", ASPPHead(256, output_stride=32, dropout_rate=dropout_rate), num_class, 32, build_input_size)
