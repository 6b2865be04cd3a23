"""backbones/feature_extractor.py of the reference (:35-189): name -> constructor (built-ins + backbone_registry_dict),
dilation surgery to `output_stride`, build, optional weight load."""
import torch

from .. import nn
from .. import static_strings as ss
from .backbone_registry import backbone_registry_dict
from .convnext import build_dilated_convnext, convnext_large, convnext_tiny, convnext_xlarge, convnext_xxlarge
from .convnext_v2 import convnext_v2_huge, convnext_v2_large, convnext_v2_nano, convnext_v2_tiny
from .hrnet import HRNetW32, HRNetW48
from .mobilenetv2_common import MobileNetV2, build_atrous_mobilenetv2
from .resnet_common import apply_multi_grid, build_atrous_resnet, resnet50, resnet101, resnet152


def _builtin_backbones():
    d = {
        ss.CONVNEXT_TINY: convnext_tiny,
        ss.CONVNEXT_LARGE: convnext_large,
        ss.CONVNEXT_XLARGE: convnext_xlarge,
        ss.CONVNEXT_XXLARGE: convnext_xxlarge,
        ss.CONVNEXT_V2_NANO: convnext_v2_nano,
        ss.CONVNEXT_V2_TINY: convnext_v2_tiny,
        ss.CONVNEXT_V2_LARGE: convnext_v2_large,
        ss.CONVNEXT_V2_HUGE: convnext_v2_huge,
        ss.MOBILENETV2: MobileNetV2,
        ss.HRNET_W48: HRNetW48,
        ss.HRNET_W32: HRNetW32,
        ss.RESNET50: resnet50,
        ss.RESNET52: resnet50,
        ss.RESNET101: resnet101,
        ss.RESNET103: resnet101,
        ss.RESNET152: resnet152,
    }
    from .swin import swin_base_384, swin_large_384, swin_tiny_224
    from .intern_image import intern_image_small, intern_image_tiny
    from .vit import ViT16B, ViT16L

    from .moat.moat import moat0, moat1, moat2, moat3, moat4

    d.update({ss.MOAT0: moat0, ss.MOAT1: moat1, ss.MOAT2: moat2, ss.MOAT3: moat3, ss.MOAT4: moat4})      # (feature_extractor.py:106-110)
    from .eva import EVA02_large_patch14_224, EVA02_large_patch16_224, EVA02_large_patch16_512_COCO, EVA02_large_patch16_512_MV, \
        EVA02_tiny_patch_14_336

    d.update({ss.EVA02_LARGE: EVA02_large_patch16_224, ss.EVA02_LARGE_P14: EVA02_large_patch14_224, ss.EVA02_TINY: EVA02_tiny_patch_14_336,
              ss.EVA02_LARGE_COCO: EVA02_large_patch16_512_COCO, ss.EVA02_LARGE_MV: EVA02_large_patch16_512_MV})      # (feature_extractor.py:121-125)
    d.update({ss.SWIN_TINY_224: swin_tiny_224, ss.SWIN_BASE_384: swin_base_384, ss.SWIN_LARGE_384: swin_large_384,
              ss.VIT_B: ViT16B, ss.VIT_L: ViT16L, ss.INTERN_IMAGE_TINY: intern_image_tiny,
              ss.INTERN_IMAGE_SMALL: intern_image_small})
    return d


def get_backbone(name=ss.RESNET50, custom_backbone_fn=None, output_stride=32, resnet_multi_grids=[1, 2, 4], resnet_slim=True,
                 custom_resblock=None, weights_path=None, return_endpoints=False, image_shape=(1, 512, 512, 3), label_shape=None,
                 efficientnet_use_top=True, moat_use_pos_encoding=False):
    name = name.lower()
    general_kwargs = {"return_endpoints": return_endpoints}
    if ss.MOAT in name:      # :73-76
        general_kwargs.update({"use_pos_emb": moat_use_pos_encoding})
    if ss.RESNET in name:      # :58-66
        general_kwargs.update({"use_bias": False, "replace_7x7_conv": True, "slim_behaviour": resnet_slim,
                               "custom_block": custom_resblock})
    backbone_dicts = _builtin_backbones()
    backbone_dicts.update(backbone_registry_dict)
    if name not in backbone_dicts:
        raise ValueError(f"Backbone {name} currently not supported")
    if custom_backbone_fn is not None:
        backbone = custom_backbone_fn(**general_kwargs)
    else:
        backbone = backbone_dicts[name](**general_kwargs)
    if ss.RESNET in name:
        build_atrous_resnet(backbone, output_stride=output_stride)
        apply_multi_grid(backbone, block_index=-1, grids=resnet_multi_grids)
    elif name == ss.MOBILENETV2:
        build_atrous_mobilenetv2(backbone, output_stride=output_stride)
    elif ss.CONVNEXT in name:
        build_dilated_convnext(backbone, output_stride=output_stride)
    # build by shape propagation (the reference runs backbone(tf.ones(image_shape)), :153-164)
    with nn.dry_run_scope():
        dummy = torch.empty(tuple(image_shape), dtype=torch.float32, device=nn.device())
        if label_shape is None:
            backbone(dummy)
        else:
            backbone((dummy, torch.empty(tuple(label_shape), dtype=torch.float32, device=nn.device())))
    print("Built backbone with shape inputs")
    if weights_path is not None:      # :166-187; `.npz` = a Keras .h5 converted by tools/h5_to_npz.py (h5py is not part of this image)
        from ..saver import load_h5_weight_by_name
        from ..utils.keras_ops import load_h5_weight

        stem = weights_path[:-4] if weights_path.endswith(".npz") else weights_path
        if stem.endswith(".topology.h5") or stem.endswith(".topology"):
            print(f"Load backbone weights {weights_path} as H5 format (topology-based)")
            load_h5_weight(backbone, weights_path, by_name=False)
        elif stem.endswith(".h5") or weights_path.endswith(".npz"):
            print(f"Load backbone weights {weights_path} as H5 format (name-based)")
            if ss.RESNET in name:      # (:174-176) ResNets keep Keras' strict loader: by layer name, weights by position
                load_h5_weight(backbone, weights_path)
            else:
                load_h5_weight_by_name(backbone, weights_path)
        elif weights_path.endswith(".ckpt") or weights_path.endswith(".keras"):
            raise NotImplementedError(f"{weights_path}: TensorFlow checkpoint / .keras archives need TensorFlow to read; export the "
                                      "backbone as .h5 there and convert it with tools/h5_to_npz.py")
        else:
            raise ValueError(f"Weights {weights_path} not supported")
    return backbone
