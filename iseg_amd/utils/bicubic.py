"""Host-side weight matrices of tf.image.resize(method="bicubic") as backbones/vit.py:49-54 calls it (TF2 default:
half-pixel centres, no antialias -> the ResizeBicubic kernel with Keys cubic a = -0.5).  Geometry only: the arithmetic of the
resize itself runs on the GPU as two GEMMs (functional._PosEmbedResizeFn).

TensorFlow semantics restated (tensorflow/core/kernels/image/resize_bicubic_op.cc, GetWeightsAndIndices with
half_pixel_centers): src = (dst + 0.5) * in/out - 0.5 in float32; the fractional part is quantised to 1/1024 (the op uses a
1024-entry coefficient table); taps in_loc-1 .. in_loc+2; a tap that falls outside the image gets weight 0 and the remaining
weights are renormalised to sum 1."""
import numpy as np

_TABLE = 1024
_A = -0.5


def _coeff_near(x):      # |distance| <= 1
    return ((_A + 2.0) * x - (_A + 3.0)) * x * x + 1.0


def _coeff_far(x):       # 1 <= |distance| <= 2
    return ((_A * x - 5.0 * _A) * x + 8.0 * _A) * x - 4.0 * _A


def bicubic_matrix(out_size, in_size):
    """W [out_size, in_size] float32 with out = W @ in along one axis"""
    W = np.zeros((out_size, in_size), dtype=np.float32)
    scale = np.float32(in_size) / np.float32(out_size)
    for o in range(out_size):
        src = np.float32((np.float32(o) + np.float32(0.5)) * scale - np.float32(0.5))
        loc = int(np.floor(src))
        delta = np.float32(src - np.float32(loc))
        offset = int(np.rint(delta * np.float32(_TABLE)))
        x0, x1 = np.float32(offset) / np.float32(_TABLE), np.float32(_TABLE - offset) / np.float32(_TABLE)
        taps = [(loc - 1, _coeff_far(np.float32(x0 + 1.0))), (loc, _coeff_near(x0)), (loc + 1, _coeff_near(x1)),
                (loc + 2, _coeff_far(np.float32(x1 + 1.0)))]
        ws = [np.float32(w) if 0 <= i < in_size else np.float32(0.0) for i, w in taps]
        total = np.float32(sum(ws))
        if abs(total) >= 1000.0 * np.finfo(np.float32).tiny:
            ws = [np.float32(w / total) for w in ws]
        for (i, _), w in zip(taps, ws):
            if 0 <= i < in_size:
                W[o, i] += w
    return W
