"""iseg_amd -- MI355X-native drop-in for the training hot path of edwardyehuang/iSeg.

Python host code on PyTorch-ROCm tensors (device memory, streams, torch.distributed/RCCL) calling the hand-written
gfx950 kernels of libiseg_hip.so through the C ABI in include/iseg_hip.h.  Module layout mirrors the reference
(backbones.backbone_registry, layers.model_builder, core_train, ...).
"""
__version__ = "0.1.0"
