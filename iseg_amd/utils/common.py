"""utils/common.py of the reference: resize_image (:107-134), get_scaled_size (:159-188), set_random_seed (:22-29),
enable_mixed_precision (:32-64)."""
import os
import random

import numpy as np
import torch

from .. import functional as F
from .. import kernels as K
from .. import nn

DEFAULT_IMAGE_RESIZE_METHOD = "bilinear"
DEFAULT_ALIGN_CORNERS = False


def set_random_seed(seed=0):
    print('Use the random seed "{}"'.format(seed))
    nn.set_seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)


def enable_mixed_precision(use_tpu=False):
    """MI355X always supports bf16 MFMA: the reference's compute-capability probe collapses to mixed_bfloat16."""
    print("GPU supports mixed_bfloat16 !")
    nn.set_compute_dtype(torch.bfloat16)


def resize_image(images, size, method=None, name=None):
    if method is None:
        method = DEFAULT_IMAGE_RESIZE_METHOD
    if isinstance(method, str):
        method = method.lower()
    if method == "bilinear":
        return F.resize_bilinear(images, size)          # tf.image.resize -> float32 -> cast back == same dtype out
    if method == "nearest":
        if images.dtype != torch.int32:
            raise NotImplementedError("nearest resize is provided for int32 label maps")
        return K.resize_nearest_i32(images.contiguous(), int(size[0]), int(size[1]))
    if method == "bicubic":
        return F.resize_bicubic(images, size)
    raise ValueError("Not support")


def get_scaled_size(inputs, scale_rate, pad_mode=0):
    h, w = int(inputs.shape[1]), int(inputs.shape[2])
    if pad_mode == 0:
        ph, pw = h % 2, w % 2
        return [int(scale_rate * float(h - ph)) + ph, int(scale_rate * float(w - pw)) + pw]
    if pad_mode != 1:
        raise ValueError(f"Not supported pad_mode = {pad_mode}")
    th, tw = int(scale_rate * h), int(scale_rate * w)
    if th % 2 == 0 and h % 2 != 0:
        th += 1
    if tw % 2 == 0 and w % 2 != 0:
        tw += 1
    return [th, tw]
