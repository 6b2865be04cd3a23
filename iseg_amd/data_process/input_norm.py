"""data_process/input_norm.py of the reference (:7-80): every variant is an affine map per channel, out = v * scale[c] + shift[c]:
ZERO_MEAN  (2/255) v - 1;  KERAS  (v - mean) / std with the ImageNet statistics;  KERAS_SCALE  the same on v / 255."""
import torch

from .. import kernels as K
from .input_norm_types import InputNormTypes

_MEAN, _STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
_EPS = 1e-7      # keras.backend.epsilon(): the denominators are max(sqrt(std^2), epsilon)


def norm_affine(input_norm_type=InputNormTypes.ZERO_MEAN):
    """(scale[3], shift[3]) of the normalisation"""
    if input_norm_type in (None, InputNormTypes.NONE):
        return [1.0] * 3, [0.0] * 3
    if input_norm_type == InputNormTypes.ZERO_MEAN:
        return [2.0 / 255.0] * 3, [-1.0] * 3
    if input_norm_type == InputNormTypes.KERAS:
        d = [max(s, _EPS) for s in _STD]
        return [1.0 / v for v in d], [-m / v for m, v in zip(_MEAN, d)]
    if input_norm_type == InputNormTypes.KERAS_SCALE:      # x / 255 with mean / 255 and std / 255
        d = [max(s / 255.0, _EPS) for s in _STD]
        return [1.0 / (255.0 * v) for v in d], [-(m / 255.0) / v for m, v in zip(_MEAN, d)]
    raise ValueError(f"Unsupported input_norm_type: {input_norm_type}")


def normalize_input_value_range(image, input_norm_type=InputNormTypes.ZERO_MEAN):
    if input_norm_type in (None, InputNormTypes.NONE):
        return image
    scale, shift = norm_affine(input_norm_type)
    x = image.contiguous()
    if x.dtype != torch.float32:
        x = x.float()
    return K.normalize_image(x, scale, shift)
