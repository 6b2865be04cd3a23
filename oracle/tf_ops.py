"""CPU restatement of the TensorFlow/Keras op semantics that edwardyehuang/iSeg's hot path relies on.

TEST INFRASTRUCTURE ONLY.  Nothing under iseg_amd/ may import this package; only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() do, and only as the checker.

PARITY UNPINNED: the reference is Python on TensorFlow/Keras, which is not installed in the build container
(no network), and the reference ships no tests or golden vectors for this path.  The
arithmetic lives in un-vendored third-party packages (TensorFlow >= 2.10 / Keras, README.md:75,87-89 of the
reference; no lock file).  What follows restates their *published* semantics by hand; each function cites
the reference call site it stands in for.  It is pinned only by closed-form known-answer tests
(tests/test_oracle_known_answers.py) and by an independent numpy loop implementation of the trickiest ops
(oracle/np_loops.py).

All tensors are NHWC torch CPU tensors (float32 or float64); every function is differentiable through torch
autograd, which is how backward passes of the HIP kernels are checked.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------------
# padding="same"  (Keras Conv2D / DepthwiseConv2D / pooling)
# ------------------------------------------------------------------------------------------------------
def same_pad(in_size, k, s=1, d=1):
    """out = ceil(in/s); total = max((out-1)*s + (k-1)*d + 1 - in, 0); before = total//2, after = rest."""
    out = -(-in_size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - in_size, 0)
    before = total // 2
    return out, before, total - before


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def conv2d(x, kernel, bias=None, strides=1, dilation=1, padding="same", groups=1):
    """keras.layers.Conv2D: x [N,H,W,Cin], kernel [kh,kw,Cin/groups,Cout] (layers/model_builder.py:54-64)."""
    sh, sw = _pair(strides)
    dh, dw = _pair(dilation)
    kh, kw = kernel.shape[0], kernel.shape[1]
    xt = x.permute(0, 3, 1, 2)
    if padding == "same":
        _, pt, pb = same_pad(x.shape[1], kh, sh, dh)
        _, pl, pr = same_pad(x.shape[2], kw, sw, dw)
        xt = F.pad(xt, (pl, pr, pt, pb))
    elif padding != "valid":
        raise ValueError(padding)
    w = kernel.permute(3, 2, 0, 1)
    y = F.conv2d(xt, w, bias, stride=(sh, sw), dilation=(dh, dw), groups=groups)
    return y.permute(0, 2, 3, 1)


def depthwise_conv2d(x, kernel, bias=None, strides=1, dilation=1, padding="same"):
    """keras.layers.DepthwiseConv2D: kernel [kh,kw,C,1] (backbones/convnext.py:25)."""
    C = x.shape[-1]
    k = kernel.reshape(kernel.shape[0], kernel.shape[1], 1, C)
    return conv2d(x, k, bias, strides, dilation, padding, groups=C)


def dense(x, kernel, bias=None):
    """keras.layers.Dense on the last axis, kernel [in,out] (backbones/convnext.py:29-30)."""
    y = x @ kernel
    return y if bias is None else y + bias


def gelu(x):
    """keras.activations.gelu(approximate=False): 0.5 x (1 + erf(x/sqrt2))  (backbones/convnext.py:53)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, gamma, beta, eps):
    """keras.layers.LayerNormalization(axis=-1): biased variance (backbones/convnext.py:27)."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def batch_norm_train(x, gamma, beta, eps, stats=None):
    """BatchNormalization(synchronized=True), training: moments from sum / sum of squares / count
    (layers/keras3/bn.py:10-73, layers/syncbn.py:70-119).  `stats` = (sum, sumsq, count) overrides the local
    moments (used to emulate the cross-replica all-reduce).  Returns y, mean, var(biased)."""
    red = tuple(range(x.dim() - 1))
    if stats is None:
        n = 1
        for a in red:
            n *= x.shape[a]
        s1, s2 = x.sum(red), (x * x).sum(red)
    else:
        s1, s2, n = stats
    mean = s1 / n
    var = s2 / n - mean * mean
    y = (x - mean) * torch.rsqrt(var + eps) * gamma + beta
    return y, mean, var


def batch_norm_infer(x, gamma, beta, moving_mean, moving_var, eps):
    return (x - moving_mean) * torch.rsqrt(moving_var + eps) * gamma + beta


def moving_update(moving, batch, momentum):
    """moving <- moving*momentum + batch*(1-momentum)  (keras BatchNormalization)."""
    return moving * momentum + batch * (1.0 - momentum)


# ------------------------------------------------------------------------------------------------------
# tf.image.resize (v2, half_pixel_centers=True, antialias=False)   -- utils/common.py:107-134
# ------------------------------------------------------------------------------------------------------
def _interp_weights(out_size, in_size, dtype, align_corners=False):
    # TF computes these in float32 (compute_interpolation_weights); keep float32 so ties resolve identically
    dst = np.arange(out_size, dtype=np.float32)
    if align_corners:      # tf.compat.v1.image.resize(..., align_corners=True): CalculateResizeScale = (in-1)/(out-1), LegacyScaler = dst*scale
        scale = np.float32(in_size - 1) / np.float32(out_size - 1) if out_size > 1 else np.float32(0.0)
        src = dst * scale
    else:
        scale = np.float32(in_size) / np.float32(out_size)
        src = (dst + np.float32(0.5)) * scale - np.float32(0.5)
    fl = np.floor(src)
    lo = np.maximum(fl, 0).astype(np.int64)
    hi = np.minimum(np.ceil(src), in_size - 1).astype(np.int64)
    t = (src - fl).astype(np.float32)
    return torch.from_numpy(lo), torch.from_numpy(hi), torch.from_numpy(t).to(dtype)


def resize_bilinear(x, size, align_corners=False):
    """value = top + (bottom-top)*ty, top = tl + (tr-tl)*tx; output float (TF returns float32).  align_corners: the legacy
    tf.compat.v1.image.resize coordinates backbones/hrnet.py:303-304,523-524 asks for."""
    Ho, Wo = size
    N, Hi, Wi, C = x.shape
    ylo, yhi, ty = _interp_weights(Ho, Hi, x.dtype, align_corners)
    xlo, xhi, tx = _interp_weights(Wo, Wi, x.dtype, align_corners)
    top_rows, bot_rows = x[:, ylo], x[:, yhi]
    tx = tx.view(1, 1, Wo, 1)
    ty = ty.view(1, Ho, 1, 1)
    top = top_rows[:, :, xlo] + (top_rows[:, :, xhi] - top_rows[:, :, xlo]) * tx
    bot = bot_rows[:, :, xlo] + (bot_rows[:, :, xhi] - bot_rows[:, :, xlo]) * tx
    return top + (bot - top) * ty


def resize_nearest(x, size):
    """src = min(floor((dst+0.5)*in/out), in-1)."""
    Ho, Wo = size
    N, Hi, Wi, C = x.shape

    def idx(o, i):
        scale = np.float32(i) / np.float32(o)
        d = np.arange(o, dtype=np.float32)
        return torch.from_numpy(np.minimum(np.floor((d + np.float32(0.5)) * scale), i - 1).astype(np.int64))

    return x[:, idx(Ho, Hi)][:, :, idx(Wo, Wi)]


# ------------------------------------------------------------------------------------------------------
# loss / metric  -- losses/catecrossentropy_ignore_label.py:44-88, metrics/*.py
# ------------------------------------------------------------------------------------------------------
def softmax_ce_ignore(y_true, logits, num_class=21, ignore_label=255, class_weights=None):
    """weighted_loss: per-position loss [N*H*W] (Reduction.NONE); Keras' wrapper then takes the mean over ALL
    positions, ignored ones included -- callers do `.mean()`."""
    z = logits.reshape(-1, num_class)
    y = y_true.reshape(-1).to(torch.int64)
    w = (y != ignore_label).to(z.dtype)
    if ignore_label == 0:
        y = y - 1
    in_range = (y >= 0) & (y < num_class)
    yc = y.clamp(0, num_class - 1)
    onehot = F.one_hot(yc, num_class).to(z.dtype) * in_range.unsqueeze(-1).to(z.dtype)  # tf.one_hot: zero row
    if class_weights is not None and len(class_weights) > 0:
        cw = torch.as_tensor(class_weights, dtype=z.dtype)
        w = w * (onehot * cw.unsqueeze(0)).sum(-1)
    logp = z - torch.logsumexp(z, dim=-1, keepdim=True)
    return -(onehot * logp).sum(-1) * w


def softmax_focal_ce_ignore(y_true, logits, num_class=21, ignore_label=255, class_weights=None, alpha=0.25, gamma=2.0):
    """use_focal_loss branch of losses/catecrossentropy_ignore_label.py:27-37: keras.losses.CategoricalFocalCrossentropy(alpha, gamma,
    from_logits=True, reduction=NONE) applied to the same one-hot rows and sample weights.  Keras' formula (the reference's
    losses/categorical_focal_crossentropy_loss.py is a copy of it): softmax -> renormalise -> clip to [1e-7, 1 - 1e-7] ->
    sum_c alpha * (1 - p_c)^gamma * (-onehot_c * log p_c)."""
    z = logits.reshape(-1, num_class)
    y = y_true.reshape(-1).to(torch.int64)
    w = (y != ignore_label).to(z.dtype)
    if ignore_label == 0:
        y = y - 1
    in_range = (y >= 0) & (y < num_class)
    yc = y.clamp(0, num_class - 1)
    onehot = F.one_hot(yc, num_class).to(z.dtype) * in_range.unsqueeze(-1).to(z.dtype)
    if class_weights is not None and len(class_weights) > 0:
        cw = torch.as_tensor(class_weights, dtype=z.dtype)
        w = w * (onehot * cw.unsqueeze(0)).sum(-1)
    p = torch.softmax(z, dim=-1)
    p = p / p.sum(-1, keepdim=True)
    p = torch.clamp(p, 1e-7, 1.0 - 1e-7)
    cce = -onehot * torch.log(p)
    return (alpha * torch.pow(1.0 - p, gamma) * cce).sum(-1) * w


def argmax_first(logits):
    """tf.argmax: first maximal index."""
    z = logits.detach().cpu().numpy()
    return torch.from_numpy(np.argmax(z, axis=-1).astype(np.int64))


def confusion_matrix(labels, preds, num_class, ignore_label):
    """metrics/seg_metric_wrapper.py:89-102 + metrics/confusion_matrix.py:65-143 (weights 0 for ignored)."""
    y = labels.reshape(-1).to(torch.int64)
    p = preds.reshape(-1).to(torch.int64)
    keep = y != ignore_label
    cm = torch.zeros(num_class, num_class, dtype=torch.float64)
    idx = y[keep] * num_class + p[keep]
    cm.view(-1).index_add_(0, idx, torch.ones(idx.numel(), dtype=torch.float64))
    return cm


def per_class_iou(cm):
    """metrics/mean_iou.py:59-77: iou_c = cm_cc / (row_c + col_c - cm_cc); classes with zero denominator drop out
    of the mean."""
    tp = cm.diag()
    denom = cm.sum(0) + cm.sum(1) - tp
    valid = denom > 0
    iou = torch.where(valid, tp / torch.where(valid, denom, torch.ones_like(denom)), torch.zeros_like(denom))
    n_valid = valid.sum()
    miou = iou.sum() / n_valid if n_valid > 0 else torch.tensor(0.0, dtype=cm.dtype)
    return iou, miou


# ------------------------------------------------------------------------------------------------------
# regularisation
# ------------------------------------------------------------------------------------------------------
def drop_path(x, keep_mask_scaled):
    """utils/drops.py:8-22 with the per-sample factor floor(keep+u)/keep given explicitly (RNG streams differ)."""
    shape = (x.shape[0],) + (1,) * (x.dim() - 1)
    return x * keep_mask_scaled.reshape(shape)


# ------------------------------------------------------------------------------------------------------
# optimisers and schedules  -- optimizers/modern/adamw.py:13-74, optimizers/modern/sgd.py:12-51,
#                              optimizers/polydecay.py:44-76
# ------------------------------------------------------------------------------------------------------
def warmup_poly_decay(step, initial_lr, decay_steps, end_lr=0.0001, warmup_steps=0, warmup_lr=1e-4, power=1.0):
    max_steps = float(decay_steps) - warmup_steps
    current = min(float(step), max_steps)
    slow = warmup_lr
    adjusted = current
    if warmup_steps > 0:
        adjusted = max(adjusted - warmup_steps, 0.0)
        slow = warmup_lr + (initial_lr - warmup_lr) * current / warmup_steps
    p = adjusted / max_steps
    lr = (initial_lr - end_lr) * (1.0 - p) ** power + end_lr
    return slow if step < warmup_steps else lr


def clip_gradients(grads, clipnorm=None, global_clipnorm=None, clipvalue=None):
    """keras optimizer base `_clip_gradients` (the reference hands clipnorm / clipvalue through, core_optimizer.py:170-183):
    clipnorm -> tf.clip_by_norm per variable (g * c / max(||g||, c)); global_clipnorm -> tf.clip_by_global_norm
    (g * c * min(1 / ||all g||, 1 / c)); clipvalue -> tf.clip_by_value.  At most one is set."""
    if clipnorm is not None and clipnorm > 0:
        return [g * clipnorm / torch.clamp(torch.sqrt((g * g).sum()), min=clipnorm) for g in grads]
    if global_clipnorm is not None and global_clipnorm > 0:
        gn = torch.sqrt(sum((g * g).sum() for g in grads))
        scale = global_clipnorm * torch.minimum(1.0 / gn, torch.tensor(1.0 / global_clipnorm, dtype=gn.dtype))
        return [g * scale for g in grads]
    if clipvalue is not None and clipvalue > 0:
        return [torch.clamp(g, -clipvalue, clipvalue) for g in grads]
    return list(grads)


def scrub_nan(g):
    """AdamW_EXT._clip_gradients (optimizers/modern/adamw.py:63-74): NaN gradients become 0 before Keras' clipping"""
    return torch.where(torch.isnan(g), torch.zeros_like(g), g)


def adamw_step(w, g, m, v, step, lr, lr_mult=1.0, wd=0.0, beta1=0.9, beta2=0.999, eps=1e-7, vhat=None):
    """One AdamW_EXT.update_step preceded by Keras' decoupled decay; `step` is 1-based (iterations+1).  vhat (amsgrad, adamw.py:54-57):
    v_hat = max(v_hat, v) takes v's place in the denominator and is returned as a fourth value."""
    g = scrub_nan(g)
    w = w - w * wd * lr
    m = m + (g - m) * (1 - beta1)
    v = v + (g * g - v) * (1 - beta2)
    alpha = lr * lr_mult * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    if vhat is not None:
        vhat = torch.maximum(vhat, v)
        return w - (m * alpha) / (torch.sqrt(vhat) + eps), m, v, vhat
    w = w - (m * alpha) / (torch.sqrt(v) + eps)
    return w, m, v


def sgd_step(w, g, m, lr, lr_mult=1.0, momentum=0.9, l2=0.0, nesterov=False):
    """SGD_EXT.update_step (optimizers/modern/sgd.py:38-51); the keras l2 regulariser of set_weight_decay adds 2 l2 w to the gradient.
    SGD_EXT does not scrub NaN gradients (only AdamW_EXT overrides _clip_gradients)."""
    g = g + 2.0 * l2 * w
    m = -g * lr * lr_mult + m * momentum
    if nesterov:
        return w + (-g * lr * lr_mult + m * momentum), m
    return w + m, m


# ------------------------------------------------------------------------------------------------------
# sliding-window tiling  -- utils/sliding_window_inference_utils.py:16-32
# ------------------------------------------------------------------------------------------------------
def sliding_start_indexs(length, crop):
    stride = int(2.0 / 3.0 * crop)
    times = (length - crop) // stride + 1
    idx = [stride * i for i in range(times)]
    if length - (times - 1) * stride > crop:
        idx.append(length - crop)
    return idx


# ------------------------------------------------------------------------------------------------------
# utils/op_utils.py:43-60  replace_nan / replace_inf / replace_nan_or_inf   (layers/fpn.py:52)
# ------------------------------------------------------------------------------------------------------
def replace_nan_or_inf(x, nan_value=0.0):
    """tf.where(is_nan(x), value, x) then tf.clip_by_value(x, min, max) with min / max taken over the tensor whose +-inf
    entries were replaced by 0.  The clip bounds are treated as constants for the gradient (they only move when an inf
    is present)."""
    x = torch.where(torch.isnan(x), torch.full_like(x, nan_value), x)
    fin = torch.where(torch.isinf(x), torch.zeros_like(x), x).detach()
    return torch.clamp(x, min=fin.min().item(), max=fin.max().item())   # gradient passes wherever min <= x <= max (tf.clip_by_value)


# ------------------------------------------------------------------------------------------------------
# layers/groupnorm.py:148-207 GroupNormalization, layers/rmsnorm.py:22-29 RMSNormalization
# ------------------------------------------------------------------------------------------------------
def group_norm(x, gamma, beta, groups, eps=1e-3):
    """reshape [N,H,W,G,C/G]; tf.nn.moments over (H, W, C/G) (biased variance); tf.nn.batch_normalization."""
    N, H, W, C = x.shape
    xg = x.reshape(N, H, W, groups, C // groups)
    mean = xg.mean(dim=(1, 2, 4), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 2, 4), keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + eps)).reshape(N, H, W, C)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    return y


def rms_norm(x, scale, eps=1e-6):
    var = (x * x).mean(dim=-1, keepdim=True)
    return x * (1.0 / torch.sqrt(var + eps)) * (1.0 + scale)


def grn(x, gamma, beta, eps=1e-6):
    """GlobalResponseNormlizationLayer.call (backbones/convnext_v2.py:45-60), x [N,H,W,C]; gamma, beta broadcast as [1,1,1,C]:
    gx = (sum_{h,w} x^2 + eps)^0.5,  nx = gx / (mean_c gx + eps),  out = gamma * (x * nx) + beta + x"""
    gx = torch.pow((x * x).sum(dim=(1, 2), keepdim=True) + eps, 0.5)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + eps)
    return gamma.reshape(1, 1, 1, -1) * (x * nx) + beta.reshape(1, 1, 1, -1) + x


# ------------------------------------------------------------------------------------------------------
# pooling with padding="SAME": keras MaxPooling2D (backbones/resnet_common.py:215-217), tf.nn.avg_pool2d
# (backbones/resnet_blocks.py:182-186; padded cells are excluded from the divisor)
# ------------------------------------------------------------------------------------------------------
def _pool_same_pads(H, W, k, s):
    kh, kw = _pair(k)
    sh, sw = _pair(s)
    _, pt, pb = same_pad(H, kh, sh)
    _, pl, pr = same_pad(W, kw, sw)
    return (kh, kw), (sh, sw), (pl, pr, pt, pb)


def max_pool_same(x, k, s):
    (kh, kw), (sh, sw), pads = _pool_same_pads(x.shape[1], x.shape[2], k, s)
    xp = F.pad(x.permute(0, 3, 1, 2), pads, value=float("-inf"))
    return F.max_pool2d(xp, (kh, kw), (sh, sw)).permute(0, 2, 3, 1)


def avg_pool_same(x, k, s):
    (kh, kw), (sh, sw), pads = _pool_same_pads(x.shape[1], x.shape[2], k, s)
    xp = F.pad(x.permute(0, 3, 1, 2), pads)
    ones = F.pad(torch.ones_like(x[:1, :, :, :1]).permute(0, 3, 1, 2), pads)
    tot = F.avg_pool2d(xp, (kh, kw), (sh, sw), divisor_override=1)
    cnt = F.avg_pool2d(ones, (kh, kw), (sh, sw), divisor_override=1)
    return (tot / cnt).permute(0, 2, 3, 1)


# ------------------------------------------------------------------------------------------------------
# tf.image.resize(method="bicubic") (backbones/vit.py:49-54): TF2 default = ResizeBicubic, half-pixel centres, Keys a=-0.5.
# Restated from tensorflow/core/kernels/image/resize_bicubic_op.cc (GetWeightsAndIndices, half_pixel_centers branch):
# float32 source coordinate, fraction quantised to 1/1024 (coefficient table), out-of-image taps dropped, weights renormalised.
# ------------------------------------------------------------------------------------------------------
def _bicubic_taps(out_size, in_size):
    a = -0.5
    taps = []
    scale = np.float32(in_size) / np.float32(out_size)
    for o in range(out_size):
        src = np.float32((np.float32(o) + np.float32(0.5)) * scale - np.float32(0.5))
        loc = math.floor(float(src))
        delta = np.float32(src - np.float32(loc))
        off = int(np.rint(delta * np.float32(1024)))

        def near(x):
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0

        def far(x):
            return ((a * x - 5.0 * a) * x + 8.0 * a) * x - 4.0 * a

        x0, x1 = off / 1024.0, (1024 - off) / 1024.0
        cand = [(loc - 1, far(x0 + 1.0)), (loc, near(x0)), (loc + 1, near(x1)), (loc + 2, far(x1 + 1.0))]
        cand = [(i, w if 0 <= i < in_size else 0.0) for i, w in cand]
        tot = sum(w for _, w in cand)
        taps.append([(min(max(i, 0), in_size - 1), w / tot) for i, w in cand])
    return taps


def resize_bicubic(x, size):
    """x [N,H,W,C] -> [N,size[0],size[1],C] (differentiable)"""
    N, H, W, C = x.shape
    ty, tx = _bicubic_taps(size[0], H), _bicubic_taps(size[1], W)
    rows = []
    for oy in range(size[0]):
        r = sum(x[:, i] * w for i, w in ty[oy])
        rows.append(r)
    t = torch.stack(rows, dim=1)                     # [N, Ho, W, C]
    cols = []
    for ox in range(size[1]):
        cols.append(sum(t[:, :, i] * w for i, w in tx[ox]))
    return torch.stack(cols, dim=2)


# ------------------------------------------------------------------------------------------------------
# attention cores
# ------------------------------------------------------------------------------------------------------
def keras_mha_self(x, wq, bq, wk, bk, wv, bv, wo, bo):
    """keras.layers.MultiHeadAttention(x, x) (backbones/vit.py:142-147,166): kernels [C,heads,d], output kernel [heads,d,C]"""
    dk = wq.shape[-1]
    q = torch.einsum("btc,chd->bthd", x, wq) + bq
    k = torch.einsum("btc,chd->bthd", x, wk) + bk
    v = torch.einsum("btc,chd->bthd", x, wv) + bv
    q = q * (float(dk) ** -0.5)
    p = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q, k), dim=-1)
    ctx = torch.einsum("bhqk,bkhd->bqhd", p, v)
    return torch.einsum("bqhd,hdc->bqc", ctx, wo) + bo


def mhsa_core(q, k, v, heads, apply_scale=True, eps=1e-7, attention_mask=None, return_attention_map=False):
    """layers/multihead_self_attention.py:106-150 on finite inputs: per-head softmax(q k^T / sqrt(d)), clip, context; attention_mask
    (1 = attend) enters as (1 - mask) * -1e9 before the softmax (utils/op_utils.py:24-38 safed_softmax)"""
    N, H, W, Cq = q.shape
    Cv = v.shape[-1]
    qh = q.reshape(N, H * W, heads, Cq // heads).permute(0, 2, 1, 3)
    kh = k.reshape(N, H * W, heads, Cq // heads).permute(0, 2, 3, 1)
    vh = v.reshape(N, H * W, heads, Cv // heads).permute(0, 2, 1, 3)
    a = qh @ kh
    if apply_scale:
        a = a / math.sqrt(Cq // heads)
    if attention_mask is not None:
        m = attention_mask.to(a.dtype)
        while m.dim() < 4:
            m = m[None] if m.dim() == 2 else m[:, None]
        a = a + (1.0 - m) * -1e9
    a = torch.softmax(a, dim=-1)
    a = torch.clamp(a, eps, 1.0 - eps)
    out = (a @ vh).permute(0, 2, 1, 3).reshape(N, H, W, Cv)
    return (out, a) if return_attention_map else out


def axial_attention_core(q, k, v, heads, apply_scale=True, eps=1e-7):
    """layers/multihead_axial_attention.py:84-146 on finite inputs: per head a column map softmax(q k^T / sqrt(d)) over H ([N, heads, W, H, H], :95-99)
    and a row map over W ([N, heads, H, W, W], :101-105), both clipped to [eps, 1 - eps] (:125-126); the value is mixed along H, then along W
    (:130-135); heads end up channel-minor (:137-139: [N, H, W, C / heads, heads] flattened)"""
    N, H, W, Cq = q.shape
    Cv = v.shape[-1]
    dq, dv = Cq // heads, Cv // heads
    qh = q.reshape(N, H, W, heads, dq)
    kh = k.reshape(N, H, W, heads, dq)
    v_map = qh.permute(0, 3, 2, 1, 4) @ kh.permute(0, 3, 2, 4, 1)      # [N, heads, W, H, H]
    u_map = qh.permute(0, 3, 1, 2, 4) @ kh.permute(0, 3, 1, 4, 2)      # [N, heads, H, W, W]
    if apply_scale:
        v_map = v_map / math.sqrt(dq)
        u_map = u_map / math.sqrt(dq)
    v_map = torch.clamp(torch.softmax(v_map, dim=-1), eps, 1.0 - eps)
    u_map = torch.clamp(torch.softmax(u_map, dim=-1), eps, 1.0 - eps)
    x = v.reshape(N, H, W, heads, dv).permute(0, 3, 2, 1, 4)             # [N, heads, W, H, C]
    x = v_map @ x
    x = x.permute(0, 1, 3, 2, 4)                                         # [N, heads, H, W, C]
    x = u_map @ x
    return x.permute(0, 2, 3, 4, 1).reshape(N, H, W, dv * heads)


# ------------------------------------------------------------------------------------------------------
# layers/dcn_v3/op.py:16-109 + utils.py:14-209, transcribed op for op (tensor form), quirks included:
# ref is stacked [y, x] while grid / offset / the sampler use [x, y]; weights come from the clipped corners.
# ------------------------------------------------------------------------------------------------------
def dcnv3_op(x, offset, mask, kernel_size=(3, 3), strides=(1, 1), padding="SAME", dilation_rate=(1, 1), groups=4, group_channels=16,
             offset_scale=1.0):
    kh, kw = kernel_size
    dh, dw = dilation_rate
    sh, sw = strides
    ph, pw = (kh // 2, kw // 2) if padding.upper() == "SAME" else (0, 0)
    x = F.pad(x, (0, 0, pw, pw, ph, ph))
    N, Hin, Win, C = x.shape
    _, Ho, Wo, _ = offset.shape
    dt_ = x.dtype
    # get_reference_points (utils.py:14-58)
    H_out = (Hin - (dh * (kh - 1) + 1)) // sh + 1
    W_out = (Win - (dw * (kw - 1) + 1)) // sw + 1
    y_start = (dh * (kh - 1)) // 2 + 0.5
    x_start = (dw * (kw - 1)) // 2 + 0.5
    ys = torch.linspace(y_start, y_start + (H_out - 1) * sh, H_out, dtype=dt_) / Hin
    xs = torch.linspace(x_start, x_start + (W_out - 1) * sw, W_out, dtype=dt_) / Win
    ref_y, ref_x = torch.meshgrid(ys, xs, indexing="ij")
    ref = torch.stack([ref_y.reshape(-1), ref_x.reshape(-1)], dim=-1).reshape(1, H_out, W_out, 1, 2)
    # generate_dilation_grids (utils.py:65-103)
    lx = torch.linspace(-((dw * (kw - 1)) // 2), -((dw * (kw - 1)) // 2) + (kw - 1) * dw, kw, dtype=dt_)
    ly = torch.linspace(-((dh * (kh - 1)) // 2), -((dh * (kh - 1)) // 2) + (kh - 1) * dh, kh, dtype=dt_)
    gx, gy = torch.meshgrid(lx, ly, indexing="ij")
    grid = torch.stack([gx / Win, gy / Hin], dim=-1).reshape(-1, 1, 2).repeat(1, groups, 1).permute(1, 0, 2)
    grid = grid.reshape(1, 1, 1, groups * kh * kw, 2)
    P_ = kh * kw
    spatial_norm = torch.tensor([Win, Hin], dtype=dt_).reshape(1, 1, 1, 2).repeat(1, 1, 1, groups * P_)
    loc = (ref + grid * offset_scale).reshape(1, Ho, Wo, groups * P_ * 2)
    loc = loc + offset * offset_scale / spatial_norm
    grids = 2 * loc - 1
    xg = x.reshape(N, Hin, Win, groups, group_channels).permute(0, 3, 1, 2, 4).reshape(N * groups, Hin, Win, group_channels)
    grids = grids.reshape(N, Ho * Wo, groups, P_, 2).permute(3, 0, 2, 1, 4).reshape(P_, N * groups, Ho * Wo, 2)
    m = mask.reshape(N, Ho * Wo, groups, P_).permute(3, 0, 2, 1).reshape(P_, N * groups, Ho * Wo, 1)
    # dcnv3_bilinear_sampler (utils.py:110-209)
    max_y, max_x = Hin - 1, Win - 1
    gx_, gy_ = grids[..., 0], grids[..., 1]
    px = 0.5 * ((gx_ + 1.0) * float(max_x - 1))
    py = 0.5 * ((gy_ + 1.0) * float(max_y - 1))
    x0 = torch.floor(px).long()
    y0 = torch.floor(py).long()
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = x0.clamp(0, max_x), x1.clamp(0, max_x)
    y0, y1 = y0.clamp(0, max_y), y1.clamp(0, max_y)
    dx0, dx1 = px - x0.to(dt_), x1.to(dt_) - px
    dy0, dy1 = py - y0.to(dt_), y1.to(dt_) - py
    wa, wb, wc, wd = dx1 * dy1, dx1 * dy0, dx0 * dy1, dx0 * dy0
    B = N * groups
    bidx = torch.arange(B).reshape(1, B, 1).expand(P_, B, Ho * Wo)
    out = 0
    for (yy, xx, ww) in ((y0, x0, wa), (y1, x0, wb), (y0, x1, wc), (y1, x1, wd)):
        out = out + xg[bidx, yy, xx] * ww.unsqueeze(-1)            # [P, B, HW, Cg]
    out = (out * m).sum(dim=0)                                      # [B, HW, Cg]
    return out.reshape(N, groups, Ho, Wo, group_channels).permute(0, 2, 3, 1, 4).reshape(N, Ho, Wo, groups * group_channels)


# ---- photometric augmentations (data_process/augments/random_{brightness,contrast,saturation,hue}_augment.py) -------------------------------
def adjust_contrast(x, factor):
    """tf.image.adjust_contrast on a float image [H, W, C]: (x - mean_hw) * factor + mean_hw, per channel"""
    x = np.asarray(x, dtype=np.float64)
    mean = x.mean(axis=(0, 1), keepdims=True)
    return (x - mean) * factor + mean


def _per_pixel_hsv(x, fn):
    """run fn(h, s, v) -> (h, s, v) on every pixel through the standard library's colorsys (an implementation independent of the kernel's);
    colorsys wants v in any scale but divides by it, so the pixels go through on their own scale like tf's float kernels"""
    import colorsys

    x = np.asarray(x, dtype=np.float64)
    out = np.empty_like(x)
    for idx in np.ndindex(x.shape[:-1]):
        r, g, b = x[idx]
        mx = max(r, g, b)
        if mx <= 0.0:                       # tf's rgb_to_hsv: s = 0 when v <= 0, hue 0
            h, s, v = 0.0, 0.0, mx
        else:
            h, s, v = colorsys.rgb_to_hsv(r, g, b)
        h, s, v = fn(h, s, v)
        out[idx] = colorsys.hsv_to_rgb(h, s, v)
    return out


def adjust_saturation(x, factor):
    """tf.image.adjust_saturation (float image, no range conversion): S * factor clipped to [0, 1]"""
    return _per_pixel_hsv(x, lambda h, s, v: (h, min(max(s * factor, 0.0), 1.0), v))


def adjust_hue(x, delta):
    """tf.image.adjust_hue: H + delta wrapped into [0, 1)"""
    return _per_pixel_hsv(x, lambda h, s, v: ((h + delta) % 1.0, s, v))


def photometric_sequence(x, brightness_delta=0.0, contrast=1.0, saturation=1.0, hue=0.0, distortions=False):
    """RandomBrightnessAugment (:12-28: + delta, clip [0, 256]) then RandomPhotoMetricDistortions.contrast_first_forward
    (random_photo_metric_distortions.py:15-37: contrast -> saturation -> hue, clip [0, 256]) on the drawn values"""
    x = np.asarray(x, dtype=np.float64)
    x = np.clip(x + brightness_delta, 0.0, 256.0)
    if distortions:
        if contrast != 1.0:
            x = adjust_contrast(x, contrast)
        if saturation != 1.0:
            x = adjust_saturation(x, saturation)
        x = np.clip(adjust_hue(x, hue), 0.0, 256.0)       # (RandomHueAugment clips too: the same clip twice)
    return x


def dcnv2(x, offset_in, kernel, bias, offset_kernel, offset_bias, dilation=1):
    """layers/dcn_v2.py:110-262 (_forward) op for op: offset convolution, (dy, dx) + sigmoid mask per kernel point, the position and its two integer
    corners clipped to the zero-padded image [0, H + 1] x [0, W + 1], the four bilinear weights from the CLIPPED values in the order
    (y1, x1), (y1, x0), (y0, x1), (y0, x0), gather, modulate, and one product with the [kh kw C, filters] kernel."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    ks, ph, pw = kh * kw, (kh - 1) // 2, (kw - 1) // 2
    off = conv2d(offset_in, offset_kernel, offset_bias, 1, dilation, "same")                 # :114-121
    B, H, W, C = x.shape
    oyox = off[..., :2 * ks].reshape(B, H, W, ks, 2)
    mask = torch.sigmoid(off[..., 2 * ks:])                                                  # :135-137
    ys, xs = torch.arange(H, dtype=x.dtype), torch.arange(W, dtype=x.dtype)
    grid = torch.stack(torch.meshgrid(ys, xs, indexing="ij"), dim=-1).reshape(1, H, W, 1, 2)      # (y, x)
    patch = torch.stack(torch.meshgrid(torch.arange(-ph, ph + 1, dtype=x.dtype), torch.arange(-pw, pw + 1, dtype=x.dtype), indexing="ij"),
                        dim=-1).reshape(ks, 2)                                               # :98-103: row-major (ky, kx)
    g = grid + torch.tensor([ph, pw], dtype=x.dtype) + patch + oyox                          # :139-147
    hi = torch.tensor([H + 1, W + 1], dtype=x.dtype)
    lo = torch.zeros(2, dtype=x.dtype)
    f = torch.floor(g)
    i1 = torch.minimum(torch.maximum(f + 1, lo), hi)                                         # :150-156
    i0 = torch.minimum(torch.maximum(f, lo), hi)                                             # :160
    gc = torch.minimum(torch.maximum(g, lo), hi)                                             # :164
    d0, d1 = gc - i0, i1 - gc                                                                # :182-183
    wts = torch.stack([d0[..., 0] * d0[..., 1], d0[..., 0] * d1[..., 1], d1[..., 0] * d0[..., 1], d1[..., 0] * d1[..., 1]], dim=-1)      # :185-199
    xp = torch.nn.functional.pad(x, (0, 0, pw, pw, ph, ph))                                  # :201
    corners = [(i1[..., 0], i1[..., 1]), (i1[..., 0], i0[..., 1]), (i0[..., 0], i1[..., 1]), (i0[..., 0], i0[..., 1])]      # :167-173
    bidx = torch.arange(B).reshape(B, 1, 1, 1).expand(B, H, W, ks)
    col = torch.zeros((B, H, W, ks, C), dtype=x.dtype)
    for k, (cy, cx) in enumerate(corners):
        col = col + wts[..., k:k + 1] * xp[bidx, cy.long(), cx.long()]
    col = col * mask.unsqueeze(-1)
    out = col.reshape(B, H * W, ks * C) @ kernel.reshape(ks * C, -1)                         # :230-240
    out = out.reshape(B, H, W, -1)
    return out if bias is None else out + bias
