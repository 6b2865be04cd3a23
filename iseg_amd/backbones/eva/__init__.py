"""backbones/eva/ of the reference: EVA-02 (plain ViT trunk with rotary position embedding, SwiGLU / GluMlp feed-forward blocks, optional sub-LayerNorm)."""
from .eva import Eva, EVA02_large_patch14_224, EVA02_large_patch14_448, EVA02_large_patch16_224, EVA02_large_patch16_512_COCO, \
    EVA02_large_patch16_512_MV, EVA02_tiny_patch_14_336  # noqa: F401
