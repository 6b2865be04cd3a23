"""data_process/ of the reference: input normalisation and the standard training augmentations, on the device."""
from .input_norm import normalize_input_value_range, norm_affine  # noqa: F401
from .input_norm_types import InputNormTypes  # noqa: F401
from .mean_pixel import get_mean_pixel  # noqa: F401
from .pipeline import StandardAugmentationsPipeline  # noqa: F401
