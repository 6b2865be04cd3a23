"""backbones/feature_extractor.py of the reference (:35-189): name -> constructor (built-ins + backbone_registry_dict),
dilation surgery to `output_stride`, build, optional weight load."""
import torch

from .. import nn
from .. import static_strings as ss
from .backbone_registry import backbone_registry_dict
from .convnext import build_dilated_convnext, convnext_large, convnext_tiny, convnext_xlarge, convnext_xxlarge
from .resnet_common import apply_multi_grid, build_atrous_resnet, resnet50, resnet101, resnet152


def _builtin_backbones():
    d = {
        ss.CONVNEXT_TINY: convnext_tiny,
        ss.CONVNEXT_LARGE: convnext_large,
        ss.CONVNEXT_XLARGE: convnext_xlarge,
        ss.CONVNEXT_XXLARGE: convnext_xxlarge,
        ss.RESNET50: resnet50,
        ss.RESNET52: resnet50,
        ss.RESNET101: resnet101,
        ss.RESNET103: resnet101,
        ss.RESNET152: resnet152,
    }
    from .swin import swin_base_384, swin_large_384, swin_tiny_224
    from .intern_image import intern_image_small, intern_image_tiny
    from .vit import ViT16B, ViT16L

    d.update({ss.SWIN_TINY_224: swin_tiny_224, ss.SWIN_BASE_384: swin_base_384, ss.SWIN_LARGE_384: swin_large_384,
              ss.VIT_B: ViT16B, ss.VIT_L: ViT16L, ss.INTERN_IMAGE_TINY: intern_image_tiny,
              ss.INTERN_IMAGE_SMALL: intern_image_small})
    return d


def get_backbone(name=ss.RESNET50, custom_backbone_fn=None, output_stride=32, resnet_multi_grids=[1, 2, 4], resnet_slim=True,
                 custom_resblock=None, weights_path=None, return_endpoints=False, image_shape=(1, 512, 512, 3), label_shape=None,
                 efficientnet_use_top=True, moat_use_pos_encoding=False):
    name = name.lower()
    general_kwargs = {"return_endpoints": return_endpoints}
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
    if weights_path is not None:
        from ..saver import load_weights_by_name

        load_weights_by_name(backbone, weights_path)
    return backbone
