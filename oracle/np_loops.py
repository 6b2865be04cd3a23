"""Second, independent statement of the trickiest TF semantics as plain numpy loops (small inputs only).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Used to pin oracle/tf_ops.py: two implementations written
from the TF documentation in different styles must agree.
"""
import math

import numpy as np


def same_pad(in_size, k, s, d):
    out = int(math.ceil(in_size / s))
    total = max((out - 1) * s + (k - 1) * d + 1 - in_size, 0)
    return out, total // 2


def conv2d_same(x, kernel, bias=None, strides=(1, 1), dilation=(1, 1), groups=1):
    """x [N,H,W,Cin], kernel [kh,kw,Cin/groups,Cout]; Keras Conv2D(padding="same")."""
    N, H, W, Cin = x.shape
    kh, kw, cpg, Cout = kernel.shape
    sh, sw = strides
    dh, dw = dilation
    Ho, pt = same_pad(H, kh, sh, dh)
    Wo, pl = same_pad(W, kw, sw, dw)
    opg = Cout // groups
    y = np.zeros((N, Ho, Wo, Cout), dtype=np.float64)
    for n in range(N):
        for oh in range(Ho):
            for ow in range(Wo):
                for i in range(kh):
                    ih = oh * sh + i * dh - pt
                    if ih < 0 or ih >= H:
                        continue
                    for j in range(kw):
                        iw = ow * sw + j * dw - pl
                        if iw < 0 or iw >= W:
                            continue
                        for g in range(groups):
                            xi = x[n, ih, iw, g * cpg:(g + 1) * cpg].astype(np.float64)
                            y[n, oh, ow, g * opg:(g + 1) * opg] += xi @ kernel[i, j, :, g * opg:(g + 1) * opg].astype(np.float64)
    if bias is not None:
        y += bias
    return y


def resize_bilinear(x, size):
    """tf.image.resize(bilinear, half-pixel centres, no antialias); index math in float32 like TF."""
    N, Hi, Wi, C = x.shape
    Ho, Wo = size
    y = np.zeros((N, Ho, Wo, C), dtype=np.float64)
    sy = np.float32(Hi) / np.float32(Ho)
    sx = np.float32(Wi) / np.float32(Wo)
    for oy in range(Ho):
        fy = (np.float32(oy) + np.float32(0.5)) * sy - np.float32(0.5)
        y0 = max(int(np.floor(fy)), 0)
        y1 = min(int(np.ceil(fy)), Hi - 1)
        ty = float(fy - np.floor(fy))
        for ox in range(Wo):
            fx = (np.float32(ox) + np.float32(0.5)) * sx - np.float32(0.5)
            x0 = max(int(np.floor(fx)), 0)
            x1 = min(int(np.ceil(fx)), Wi - 1)
            tx = float(fx - np.floor(fx))
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * tx
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * tx
            y[:, oy, ox] = top + (bot - top) * ty
    return y


def layer_norm(x, gamma, beta, eps):
    y = np.zeros_like(x, dtype=np.float64)
    flat = x.reshape(-1, x.shape[-1]).astype(np.float64)
    out = y.reshape(-1, x.shape[-1])
    for r in range(flat.shape[0]):
        mu = flat[r].sum() / flat.shape[1]
        var = ((flat[r] - mu) ** 2).sum() / flat.shape[1]
        out[r] = (flat[r] - mu) / math.sqrt(var + eps) * gamma + beta
    return y


def softmax_ce_ignore(y_true, logits, num_class, ignore_label, class_weights=None):
    z = logits.reshape(-1, num_class).astype(np.float64)
    y = y_true.reshape(-1)
    out = np.zeros(z.shape[0], dtype=np.float64)
    for p in range(z.shape[0]):
        lab = int(y[p])
        if lab == ignore_label:
            continue
        if ignore_label == 0:
            lab -= 1
        if lab < 0 or lab >= num_class:
            continue
        m = z[p].max()
        lse = m + math.log(np.exp(z[p] - m).sum())
        w = 1.0 if class_weights is None else float(class_weights[lab])
        out[p] = w * (lse - z[p, lab])
    return out
