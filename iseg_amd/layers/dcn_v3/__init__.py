from .dcn_v3 import DeformableConvolutionV3  # noqa: F401
