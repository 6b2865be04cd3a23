"""Thin tensor-level wrappers over the C ABI (include/iseg_hip.h).

torch is used here only for device memory (torch.empty), dtype tags and the current HIP stream; every FLOP
and every byte moved on the hot path happens inside libiseg_hip.so.  Nothing here falls back to torch ops.
"""
import contextlib
import ctypes as C

import torch

from . import _hip
from ._hip import F32, BF16, ACT_NONE, ACT_RELU, ACT_GELU, ACT_GELU_GRAD, ACT_RELU_GRAD, ACT_MUL_AUX, ACT_SIGMOID, ACT_SWISH, GemmArgs  # noqa: F401

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise TypeError(f"iseg_amd kernels support float32 and bfloat16 storage, got {t.dtype if isinstance(t, torch.Tensor) else t}")


def ptr(t):
    return None if t is None else t.data_ptr()


_DEVICE_INDEX = [None]


def stream():
    """raw hipStream_t of torch's current stream on this process's device.  torch.cuda.current_stream() re-derives the device
    through is_available() -> os.getenv on every call (measured 70 us per launch on the GPU box: 20 of the 41 ms of a batch-1
    ViT inference step were spent there); one process drives one GPU, so the index is resolved once."""
    idx = _DEVICE_INDEX[0]
    if idx is None:
        idx = _DEVICE_INDEX[0] = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _hip.HipCallError("iseg_amd kernels need device tensors (cuda:N); there is no CPU path")


# ---------------------------------------------------------------------------------------------------------
# workspace: one growable byte buffer per (device, stream); stream order makes reuse across ops safe
# ---------------------------------------------------------------------------------------------------------
_WS = {}


def workspace(nbytes, device):
    if nbytes <= 0:
        return None, 0
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        size = int(nbytes * 1.25) + 1024
        if torch.cuda.is_current_stream_capturing():
            if buf is not None:      # kernels already captured hold the old pointer: it must not go back to the pool
                raise _hip.HipCallError("workspace growth during hipGraph capture; run one eager warm-up step first")
            # a stream that first appears inside the capture (the side streams of functional.parallel_branches / _SideQueue): give it what
            # the warmed-up streams of this device ended up needing, so that it never has to grow while the capture goes on
            size = max([size] + [b.numel() for (d, _), b in _WS.items() if d == key[0]])
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf, buf.numel()


# ---------------------------------------------------------------------------------------------------------
# optional per-launch timing with HIP events on the launch stream (bench.py's live roofline measurement)
# ---------------------------------------------------------------------------------------------------------
KERNEL_TIMER = [None]


class KernelTimer:
    """records a (start, stop) event pair around every instrumented launch, grouped by a shape key; no host sync until report()"""

    def __init__(self, only=None, every=1):
        self.records = {}
        self.seen = {}          # key -> launches seen (timed or not)
        self._cur = None
        self.only = only        # None: every instrumented launch; else the one shape key (or a set / list of keys) to time
        self._only = None if only is None else (set(only) if isinstance(only, (set, list, frozenset)) else {only})
        self.every = max(1, int(every))      # time every k-th matching launch: an event pair costs the stream a few microseconds of idle time

    def begin(self, key):
        if self._only is not None and key not in self._only:
            self._cur = None
            return
        n = self.seen.get(key, 0)
        self.seen[key] = n + 1
        if n % self.every:
            self._cur = None
            return
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._cur = (key, e0)

    def end(self):
        if self._cur is None:
            return
        key, e0 = self._cur
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.setdefault(key, []).append((e0, e1))

    def report(self):
        torch.cuda.synchronize()
        out = {}
        for key, pairs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[key] = (sum(ms) / len(ms) * 1e-3, self.seen.get(key, len(ms)))     # seconds per launch (over the timed ones), launches seen
        return out


# ---------------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------------
def gemm(A, B, D, M, N, K, *, lda, ldb, ldd, a_kcontig, b_kcontig, bias=None, colscale=None, rowscale=None,
         rows_per_group=0, residual=None, ldr=0, aux=None, ldaux=0, pre_out=None, ldp=0, act=ACT_NONE, alpha=1.0,
         accumulate=False, split_k=0, a_act=ACT_NONE, colsum_out=None, colsum_accumulate=False, batch=1, batch_inner=1,
         sa=(0, 0), sb=(0, 0), sd=(0, 0), pre_deriv=False, b_group=None, bias_rowscaled=False):
    """batch > 1: problem z uses X + (z // batch_inner) * sX[0] + (z % batch_inner) * sX[1] (element strides);
    b_group=(rows, stride): rows [i*rows, (i+1)*rows) of A / D use B + i*stride (one kernel per sample; LDS-DMA path only)"""
    _require_cuda(A, B, D)
    if A.dtype != B.dtype:
        raise TypeError(f"gemm operands differ in dtype: {A.dtype} vs {B.dtype}")
    g = GemmArgs()
    g.A, g.lda, g.a_kcontig = ptr(A), lda, int(a_kcontig)
    g.B, g.ldb, g.b_kcontig = ptr(B), ldb, int(b_kcontig)
    g.D, g.ldd = ptr(D), ldd
    g.M, g.N, g.K = M, N, K
    g.in_dtype, g.out_dtype = dt(A), dt(D)
    g.bias, g.colscale, g.rowscale, g.rows_per_group = ptr(bias), ptr(colscale), ptr(rowscale), rows_per_group
    g.residual, g.ldr = ptr(residual), ldr
    g.aux, g.ldaux = ptr(aux), ldaux
    g.pre_out, g.ldp = ptr(pre_out), ldp
    g.act, g.alpha, g.accumulate, g.split_k, g.a_act = act, alpha, int(accumulate), split_k, a_act
    g.colsum_out, g.colsum_accumulate = ptr(colsum_out), int(colsum_accumulate)
    g.batch, g.batch_inner = int(batch), int(batch_inner)
    g.pre_deriv = int(bool(pre_deriv))
    g.bias_rowscaled = int(bool(bias_rowscaled))
    g.b_group_rows, g.b_group_stride = (int(b_group[0]), int(b_group[1])) if b_group is not None else (0, 0)
    (g.sa_outer, g.sa_inner), (g.sb_outer, g.sb_inner), (g.sd_outer, g.sd_inner) = sa, sb, sd
    for t in (residual, aux, pre_out):
        if t is not None and t.dtype != D.dtype:
            raise TypeError("gemm residual/aux/pre_out must have the output dtype")
    for t in (bias, colscale, rowscale):
        if t is not None and t.dtype != torch.float32:
            raise TypeError("gemm bias/colscale/rowscale must be float32")
    L = _hip.lib()
    need = L.iseg_gemm_workspace_bytes(C.byref(g))
    ws, wsb = workspace(need, A.device)
    g.defer_reduce = 1 if need > 0 else 0
    timer = KERNEL_TIMER[0]
    if timer is not None:
        timer.begin(("gemm", int(a_kcontig), int(b_kcontig), int(M), int(N), int(K), int(act), pre_out is not None, residual is not None,
                     aux is not None, A.dtype, D.dtype, need > 0, int(L.iseg_gemm_variant(C.byref(g)))) + ((int(batch),) if batch > 1 else ()))
    _hip.check(L.iseg_gemm(C.byref(g), ptr(ws), wsb, stream()), "iseg_gemm")
    if timer is not None:
        timer.end()
    if need > 0:
        _hip.check(L.iseg_gemm_reduce(C.byref(g), ptr(ws), wsb, stream()), "iseg_gemm_reduce")
    return D


def dense_fwd(x2d, W, bias=None, *, act=ACT_NONE, out=None, ldd=None, out_dtype=None, pre_out=None, colscale=None,
              rowscale=None, rows_per_group=0, residual=None, a_act=ACT_NONE, pre_deriv=False):
    """x2d [M,K] (row stride x2d.stride(0)) @ W [K,N] (Keras Dense / 1x1 conv kernel).  pre_deriv: pre_out receives gelu'(pre-activation)
    (act must be GELU); the matching backward epilogue is ACT_MUL_AUX."""
    M, K = x2d.shape
    N = W.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or x2d.dtype, device=x2d.device)
    return gemm(x2d, W, out, M, N, K, lda=x2d.stride(0), ldb=W.stride(0), ldd=ldd or out.stride(0), a_kcontig=1, b_kcontig=0,
                bias=bias, act=act, pre_out=pre_out, ldp=(pre_out.stride(0) if pre_out is not None else 0), colscale=colscale,
                rowscale=rowscale, rows_per_group=rows_per_group, residual=residual,
                ldr=(residual.stride(0) if residual is not None else 0), a_act=a_act, pre_deriv=pre_deriv)


def dense_fwd_t(x2d, Wt, bias=None, *, act=ACT_NONE, out=None, pre_out=None, colscale=None, rowscale=None, rows_per_group=0, residual=None,
                pre_deriv=False, bias_rowscaled=False):
    """dense_fwd with the kernel given K-contiguous: Wt [N,K] (nn.wt): both operands K-contiguous, so the LDS-DMA GEMM serves it"""
    M, K = x2d.shape
    N = Wt.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x2d.dtype, device=x2d.device)
    return gemm(x2d, Wt, out, M, N, K, lda=x2d.stride(0), ldb=Wt.stride(0), ldd=out.stride(0), a_kcontig=1, b_kcontig=1, bias=bias, act=act,
                pre_out=pre_out, ldp=(pre_out.stride(0) if pre_out is not None else 0), colscale=colscale, rowscale=rowscale,
                rows_per_group=rows_per_group, residual=residual, ldr=(residual.stride(0) if residual is not None else 0), pre_deriv=pre_deriv,
                bias_rowscaled=bias_rowscaled)


def dense_dgrad(dy2d, W, *, out=None, act=ACT_NONE, aux=None, rowscale=None, rows_per_group=0, residual=None, accumulate=False):
    """dX [M,K] = dY [M,N] @ W[K,N]^T : W is consumed as stored (its rows are the N-contiguous reduction)."""
    M, N = dy2d.shape
    K = W.shape[0]
    if out is None:
        out = torch.empty((M, K), dtype=dy2d.dtype, device=dy2d.device)
    return gemm(dy2d, W, out, M, K, N, lda=dy2d.stride(0), ldb=W.stride(0), ldd=out.stride(0), a_kcontig=1, b_kcontig=1, act=act,
                aux=aux, ldaux=(aux.stride(0) if aux is not None else 0), rowscale=rowscale, rows_per_group=rows_per_group,
                residual=residual, ldr=(residual.stride(0) if residual is not None else 0), accumulate=accumulate)


def wgrad_can_fuse_bias(x2d):
    """the bias gradient can ride the weight-gradient GEMM (virtual ones-row = output row K): free when the last 128-row output tile has a spare
    row, one more tile row otherwise -- measured worth it from 3 tile rows on (flagship step 10.05 -> 9.94 ms when the stage-2 b1 gradients
    joined: the column-sum pass over the [M, 4C] tensor and its reduce cost more than a quarter more weight-gradient tiles)"""
    k = x2d.shape[1]
    return x2d.dtype == torch.bfloat16 and k % 8 == 0 and (k % 128 != 0 or k >= 384)


def dense_wgrad(x2d, dy2d, out, *, accumulate=True, alpha=1.0, a_act=ACT_NONE, bias_grad=None):
    """dW [K,N] (+)= X[M,K]^T @ dY[M,N]; out is fp32.  bias_grad [N] (+)= colsum(dY) if given (fused when possible)."""
    M, K = x2d.shape
    N = dy2d.shape[1]
    fuse = bias_grad is not None and wgrad_can_fuse_bias(x2d)
    if bias_grad is not None and not fuse:
        colsum(dy2d, dy2d.stride(0), 0, 1, M, N, bias_grad, accumulate=accumulate)
    return gemm(x2d, dy2d, out, K, N, M, lda=x2d.stride(0), ldb=dy2d.stride(0), ldd=out.stride(0), a_kcontig=0, b_kcontig=0,
                accumulate=accumulate, alpha=alpha, a_act=a_act, colsum_out=(bias_grad if fuse else None),
                colsum_accumulate=accumulate)


def dense_wgrad_slabs(x2d, dy2d, ones_row=True):
    """the weight-gradient product X^T dY with its bias-gradient ones-row, stopped in front of the slab sum: (slabs [n, K + 1, N] fp32, n) for a
    consumer that sums the split-K slabs while it reads them (layerscale_grads_slabs), or None when the problem is not split / cannot carry
    the ones-row"""
    M, K = x2d.shape
    N = dy2d.shape[1]
    if (ones_row and not wgrad_can_fuse_bias(x2d)) or N % 4 != 0 or x2d.dtype != torch.bfloat16:
        return None
    _require_cuda(x2d, dy2d)
    g = GemmArgs()
    dummy = torch.empty(1, dtype=torch.float32, device=x2d.device)      # (never written: the problem stops at its slabs)
    g.A, g.lda, g.a_kcontig = ptr(x2d), x2d.stride(0), 0
    g.B, g.ldb, g.b_kcontig = ptr(dy2d), dy2d.stride(0), 0
    g.D, g.ldd = ptr(dummy), N
    g.M, g.N, g.K = K, N, M
    g.in_dtype, g.out_dtype = dt(x2d), 0
    g.alpha, g.accumulate = 1.0, 0
    g.colsum_out, g.colsum_accumulate = (ptr(dummy) if ones_row else None), 0      # ones_row=False: slabs [n, K, N] (layerscale_grads_slabs srow=...)
    g.batch, g.batch_inner = 1, 1
    L = _hip.lib()
    need = L.iseg_gemm_workspace_bytes(C.byref(g))
    if need == 0:
        return None
    eff = int(L.iseg_gemm_slabs(C.byref(g)))
    slabs = torch.empty(need // 4, dtype=torch.float32, device=x2d.device)      # (its own buffer: the consumer's partials use the workspace)
    g.defer_reduce = 1
    timer = KERNEL_TIMER[0]
    if timer is not None:
        timer.begin(("gemm", 0, 0, int(K), int(N), int(M), 0, False, False, False, x2d.dtype, torch.float32, True, int(L.iseg_gemm_variant(C.byref(g)))))
    _hip.check(L.iseg_gemm(C.byref(g), ptr(slabs), need, stream()), "iseg_gemm")
    if timer is not None:
        timer.end()
    return slabs, int(eff)


def dense_wgrad_pair(g2d, dbr2d, x2d, dh2d, dW1, db1, accumulate=True, defer_second=False, ones_first=True):
    """The two weight-gradient products of an un-fused ConvNeXt block as ONE launch (csrc/gemm_dma_tn.h, round 5): Z = g^T dbr [4C, C] with its
    ones-row, stopped at its slabs for layerscale_grads_slabs, and dW1 (+)= x^T dh [C, 4C] (+ db1 from its ones-row), summed by iseg_gemm_reduce.
    Returns (slabs of Z, slab count) or None when the problems do not pair (the caller then runs them one by one)."""
    rows, K4 = g2d.shape
    Cc = dbr2d.shape[1]
    if not (wgrad_can_fuse_bias(g2d) and wgrad_can_fuse_bias(x2d)) or Cc % 4 != 0:
        return None
    _require_cuda(g2d, dbr2d, x2d, dh2d, dW1)
    dummy = torch.empty(1, dtype=torch.float32, device=g2d.device)

    def args(a, b, out, ldd, bias, acc):
        g = GemmArgs()
        g.A, g.lda, g.a_kcontig = ptr(a), a.stride(0), 0
        g.B, g.ldb, g.b_kcontig = ptr(b), b.stride(0), 0
        g.D, g.ldd = ptr(out), ldd
        g.M, g.N, g.K = a.shape[1], b.shape[1], rows
        g.in_dtype, g.out_dtype = dt(a), 0
        g.alpha, g.accumulate = 1.0, int(acc)
        g.colsum_out, g.colsum_accumulate = ptr(bias), int(acc)
        g.batch, g.batch_inner = 1, 1
        g.defer_reduce = 1
        return g

    # ones_first=False (round 6): Z carries no ones-row -- its column sums S come from layerscale_grads_slabs(srow=...) instead
    g0 = args(g2d, dbr2d, dummy, Cc, dummy if ones_first else None, False)
    g1 = args(x2d, dh2d, dW1, dW1.stride(0), db1, accumulate)
    L = _hip.lib()
    s = int(L.iseg_gemm_tn_pair_splits(C.byref(g0), C.byref(g1)))
    if s <= 1:
        return None
    g0.split_k = g1.split_k = s
    need0, need1 = L.iseg_gemm_workspace_bytes(C.byref(g0)), L.iseg_gemm_workspace_bytes(C.byref(g1))
    eff = int(L.iseg_gemm_slabs(C.byref(g0)))
    slabs0 = torch.empty(need0 // 4, dtype=torch.float32, device=g2d.device)      # (its own buffer: the consumer's partials use the workspace)
    # defer_second (round 6): the second product's slabs get their own buffer and are NOT summed here -- layerscale_grads_slabs(extra=...) sums them
    # in the launch that consumes the first product's slabs (one launch instead of two); only for the plain dense case the wide reducer covers
    n2 = (x2d.shape[1] + 1) * dh2d.shape[1]
    n0 = x2d.shape[1] * dh2d.shape[1]
    defer_second = bool(defer_second and dW1.stride(0) == dh2d.shape[1] and n2 % 4 == 0 and n0 % 4 == 0 and dW1.is_contiguous() and db1 is not None)
    if defer_second:
        ws1 = torch.empty(need1 // 4, dtype=torch.float32, device=g2d.device)
        wsb1 = need1
    else:
        ws1, wsb1 = workspace(need1, g2d.device)
    timer = KERNEL_TIMER[0]
    if timer is not None:
        # (variant 9 = the pair launch: bench.py models it as two products [4C, C] and [C, 4C] over `rows`)
        timer.begin(("gemm", 0, 0, int(K4), int(Cc), int(rows), 0, False, False, False, g2d.dtype, torch.float32, True, 9))
    _hip.check(L.iseg_gemm_tn_pair(C.byref(g0), ptr(slabs0), need0, C.byref(g1), ptr(ws1), wsb1, stream()), "iseg_gemm_tn_pair")
    if timer is not None:
        timer.end()
    if defer_second:
        eff1 = int(L.iseg_gemm_slabs(C.byref(g1)))
        return slabs0, eff, (ws1, eff1, n2, dW1, db1, n0, bool(accumulate))
    _hip.check(L.iseg_gemm_reduce(C.byref(g1), ptr(ws1), wsb1, stream()), "iseg_gemm_reduce")
    return slabs0, eff


# ---------------------------------------------------------------------------------------------------------
# norms
# ---------------------------------------------------------------------------------------------------------
def convnext_mlp_supported(C, dtype):
    """the fused ConvNeXt MLP kernels (csrc/mlp_fused.hip) cover bf16 storage at C = 96 / 192"""
    return bool(_hip.lib().iseg_convnext_mlp_supported(int(C), _DT[dtype])) if dtype in _DT else False


def convnext_mlp_prep(W1, W2, gamma, backward=True):
    """fp32 Keras kernels -> the tiled bf16 images the fused kernels stream (once per step and block); returns (fw_tiled, bw_tiled)"""
    _require_cuda(W1, W2)
    Cc = W1.shape[0]
    L = _hip.lib()
    fw = torch.empty(L.iseg_convnext_mlp_tiled_bytes(Cc, 0) // 2, dtype=torch.bfloat16, device=W1.device)
    bw = torch.empty(L.iseg_convnext_mlp_tiled_bytes(Cc, 1) // 2, dtype=torch.bfloat16, device=W1.device) if backward else None
    _hip.call("iseg_convnext_mlp_prep", ptr(W1), ptr(W2), ptr(gamma), ptr(fw), ptr(bw), Cc, stream())
    return fw, bw


def convnext_mlp_fwd(y2, fw_tiled, b1, b2, gamma, rowscale, rows_per_group, residual, out=None):
    """out = residual + rowscale[m // rows_per_group] * gamma * (gelu(y2 @ W1 + b1) @ W2 + b2); the [M, 4C] hidden tile stays on the CU"""
    _require_cuda(y2, fw_tiled, residual)
    M, Cc = y2.shape
    if out is None:
        out = torch.empty((M, Cc), dtype=y2.dtype, device=y2.device)
    _hip.call("iseg_convnext_mlp_fwd", ptr(y2), ptr(fw_tiled), ptr(b1), ptr(b2), ptr(gamma), ptr(rowscale), int(rows_per_group),
              ptr(residual), ptr(out), M, Cc, dt(y2), stream())
    return out


def convnext_mlp_bwd(y2, dbr, bw_tiled, b1):
    """backward chain with the hidden tile recomputed: returns (g = gelu(h), dh = (dbr @ (W2 gamma)^T) * gelu'(h), dy2 = dh @ W1^T)"""
    _require_cuda(y2, dbr, bw_tiled)
    M, Cc = y2.shape
    g = torch.empty((M, 4 * Cc), dtype=y2.dtype, device=y2.device)
    dh = torch.empty((M, 4 * Cc), dtype=y2.dtype, device=y2.device)
    dy2 = torch.empty((M, Cc), dtype=y2.dtype, device=y2.device)
    _hip.call("iseg_convnext_mlp_bwd", ptr(y2), ptr(dbr), ptr(bw_tiled), ptr(b1), ptr(g), ptr(dh), ptr(dy2), M, Cc, dt(y2), stream())
    return g, dh, dy2


def convnext_mlp_fwd_ln(y1, ln_gamma, ln_beta, eps, fw_tiled, b1, b2, gamma, rowscale, rows_per_group, residual):
    """convnext_mlp_fwd with LayerNorm(y1) formed while the rows are loaded: returns (out, mean, rstd); y2 is never written"""
    _require_cuda(y1, fw_tiled, residual)
    M, Cc = y1.shape
    out = torch.empty((M, Cc), dtype=y1.dtype, device=y1.device)
    mean = torch.empty(M, dtype=torch.float32, device=y1.device)
    rstd = torch.empty(M, dtype=torch.float32, device=y1.device)
    _hip.call("iseg_convnext_mlp_fwd_ln", ptr(y1), ptr(ln_gamma), ptr(ln_beta), float(eps), ptr(mean), ptr(rstd), ptr(fw_tiled), ptr(b1), ptr(b2),
              ptr(gamma), ptr(rowscale), int(rows_per_group), ptr(residual), ptr(out), M, Cc, dt(y1), stream())
    return out, mean, rstd


def convnext_mlp_bwd_data(y, dout, bw_tiled, b1, rowscale=None, rows_per_group=0, ln=None):
    """dy2 = ((rowscale * dout) @ (W2 gamma)^T * gelu'(h)) @ W1^T with the hidden tile recomputed and kept on the CU (nothing [M, 4C] is
    written); ln = (mean, rstd, ln_gamma, ln_beta) makes `y` the LayerNorm input"""
    _require_cuda(y, dout, bw_tiled)
    M, Cc = y.shape
    dy2 = torch.empty((M, Cc), dtype=y.dtype, device=y.device)
    mean, rstd, lng, lnb = ln if ln is not None else (None, None, None, None)
    _hip.call("iseg_convnext_mlp_bwd_data", ptr(y), ptr(mean), ptr(rstd), ptr(lng), ptr(lnb), ptr(dout), ptr(rowscale), int(rows_per_group),
              ptr(bw_tiled), ptr(b1), ptr(dy2), M, Cc, dt(y), stream())
    return dy2


def convnext_mlp_bwd_data_ln(y1, dout, bw_tiled, b1, ln, dln_gamma, dln_beta, rowscale=None, rows_per_group=0):
    """convnext_mlp_bwd_data carried through the LayerNorm in front of the MLP: returns the gradient of the LayerNorm INPUT y1 and adds the
    LayerNorm parameter gradients to dln_gamma / dln_beta; ln = (mean, rstd, ln_gamma, ln_beta)"""
    _require_cuda(y1, dout, bw_tiled, dln_gamma, dln_beta)
    M, Cc = y1.shape
    dy1 = torch.empty((M, Cc), dtype=y1.dtype, device=y1.device)
    mean, rstd, lng, lnb = ln
    need = _hip.lib().iseg_convnext_mlp_bwd_data_ln_workspace_bytes(M, Cc)
    ws, wsb = workspace(need, y1.device)
    _hip.call("iseg_convnext_mlp_bwd_data_ln", ptr(y1), ptr(mean), ptr(rstd), ptr(lng), ptr(lnb), ptr(dout), ptr(rowscale), int(rows_per_group),
              ptr(bw_tiled), ptr(b1), ptr(dy1), ptr(dln_gamma), ptr(dln_beta), M, Cc, dt(y1), ptr(ws), wsb, stream())
    return dy1


def convnext_mlp_wgrad(y, dout, bw_tiled, b1, W2, b2, gamma, dW1, db1, dW2, db2, dgamma, rowscale=None, rows_per_group=0, ln=None):
    """all parameter gradients of the fused MLP accumulated into dW1 / db1 / dW2 / db2 / dgamma (csrc/mlp_wgrad.hip); ln = (mean, rstd,
    ln_gamma, ln_beta) makes `y` the LayerNorm input"""
    _require_cuda(y, dout, bw_tiled)
    M, Cc = y.shape
    need = _hip.lib().iseg_convnext_mlp_wgrad_workspace_bytes(M, Cc)
    ws, wsb = workspace(need, y.device)
    mean, rstd, lng, lnb = ln if ln is not None else (None, None, None, None)
    _hip.call("iseg_convnext_mlp_wgrad", ptr(y), ptr(mean), ptr(rstd), ptr(lng), ptr(lnb), ptr(dout), ptr(rowscale), int(rows_per_group),
              ptr(bw_tiled), ptr(b1), ptr(W2), ptr(b2), ptr(gamma), ptr(dW1), ptr(db1), ptr(dW2), ptr(db2), ptr(dgamma), M, Cc, dt(y), ptr(ws),
              wsb, stream())


def layernorm_fwd(x2d, gamma, beta, eps, save_stats=True):
    _require_cuda(x2d)
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device) if save_stats else None
    _hip.call("iseg_layernorm_fwd", ptr(x2d), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, Cc, eps, dt(x2d), stream())
    return y, mean, rstd


def layernorm_bwd(dy2d, x2d, gamma, mean, rstd, dgamma, dbeta, dx_add=None, accumulate=True):
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    need = _hip.lib().iseg_layernorm_bwd_workspace_bytes(rows, Cc)
    ws, wsb = workspace(need, x2d.device)
    _hip.call("iseg_layernorm_bwd", ptr(dy2d), ptr(x2d), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dx_add), ptr(dgamma),
              ptr(dbeta), int(accumulate), rows, Cc, dt(x2d), ptr(ws), wsb, stream())
    return dx


def layernorm_post_fwd(x2d, gamma, beta, eps, colscale=None, rowscale=None, rows_per_group=0, residual=None):
    """residual + rowscale[row // rows_per_group] * colscale * LN(x2d) in one pass (iseg_layernorm_post_fwd); returns (y, mean, rstd)"""
    _require_cuda(x2d)
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    _hip.call("iseg_layernorm_post_fwd", ptr(x2d), ptr(gamma), ptr(beta), ptr(colscale), ptr(rowscale), int(rows_per_group), ptr(residual), ptr(y),
              ptr(mean), ptr(rstd), rows, Cc, eps, dt(x2d), stream())
    return y, mean, rstd


def layernorm_post_bwd(dy2d, x2d, gamma, beta, mean, rstd, dgamma, dbeta, colscale=None, dcolscale=None, rowscale=None, rows_per_group=0):
    """backward of layernorm_post_fwd: returns dx; dgamma / dbeta / dcolscale are accumulated"""
    _require_cuda(dy2d, x2d)
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    need = _hip.lib().iseg_layernorm_bwd_workspace_bytes(rows, Cc) + 2 * Cc * 4
    ws, wsb = workspace(need, x2d.device)
    _hip.call("iseg_layernorm_post_bwd", ptr(dy2d), ptr(x2d), ptr(gamma), ptr(beta), ptr(colscale), ptr(rowscale), int(rows_per_group), ptr(mean),
              ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dcolscale), rows, Cc, dt(x2d), ptr(ws), wsb, stream())
    return dx


def layernorm_gather_fwd(x2d, src_index, gamma, beta, eps):
    """y[r] = LN(x2d[src_index[r]]) or a zero row where src_index[r] < 0; mean / rstd per OUTPUT row (iseg_layernorm_gather_fwd)"""
    _require_cuda(x2d, src_index)
    rows_out, Cc = src_index.numel(), x2d.shape[1]
    y = torch.empty((rows_out, Cc), dtype=x2d.dtype, device=x2d.device)
    mean = torch.empty(rows_out, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(rows_out, dtype=torch.float32, device=x2d.device)
    _hip.call("iseg_layernorm_gather_fwd", ptr(x2d), ptr(src_index), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows_out, Cc, eps,
              dt(x2d), stream())
    return y, mean, rstd


def layernorm_gather_bwd(dy2d, dy_index, x2d, gamma, mean, rstd, dgamma, dbeta, dx_add=None, accumulate=True):
    """backward of layernorm_gather_fwd over the source rows: the gradient row / statistics of source row r are row dy_index[r] of dy2d / mean / rstd"""
    _require_cuda(dy2d, dy_index, x2d)
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    need = _hip.lib().iseg_layernorm_bwd_workspace_bytes(rows, Cc)
    ws, wsb = workspace(need, x2d.device)
    _hip.call("iseg_layernorm_gather_bwd", ptr(dy2d), ptr(dy_index), ptr(x2d), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dx_add), ptr(dgamma),
              ptr(dbeta), int(accumulate), rows, Cc, dt(x2d), ptr(ws), wsb, stream())
    return dx


def gather_rows_fma(x2d, idx, scale=None, rows_per_group=0, scale_by_source_row=False, residual=None):
    """y[r] = residual[r] + scale[g] * x2d[idx[r]] (zero where idx[r] < 0), g = (idx[r] if scale_by_source_row else r) // rows_per_group"""
    _require_cuda(x2d, idx)
    rows_out, Cc = idx.numel(), x2d.shape[1]
    y = torch.empty((rows_out, Cc), dtype=x2d.dtype, device=x2d.device)
    _hip.call("iseg_gather_rows_fma", ptr(x2d), ptr(idx), ptr(scale), int(rows_per_group), int(scale_by_source_row), ptr(residual), ptr(y), rows_out,
              Cc, dt(x2d), stream())
    return y


def bn_stats(x2d, ldx, rows, Cc, out=None):
    """packed [2C+1] = (sum, sum of squares, count); `out`: a slice of a larger message (several layers, one all-reduce)"""
    packed = out if out is not None else torch.empty(2 * Cc + 1, dtype=torch.float32, device=x2d.device)
    need = _hip.lib().iseg_bn_workspace_bytes(rows, Cc)
    ws, wsb = workspace(need, x2d.device)
    _hip.call("iseg_bn_stats", ptr(x2d), ldx, ptr(packed), rows, Cc, dt(x2d), ptr(ws), wsb, stream())
    return packed


def bn_finalize(packed, Cc, eps, momentum, moving_mean, moving_var):
    mean = torch.empty(Cc, dtype=torch.float32, device=packed.device)
    rstd = torch.empty(Cc, dtype=torch.float32, device=packed.device)
    _hip.call("iseg_bn_finalize", ptr(packed), Cc, eps, momentum, ptr(mean), ptr(rstd), ptr(moving_mean), ptr(moving_var), stream())
    return mean, rstd


def bn_finalize_apply(packed, x2d, ldx, gamma, beta, y2d, ldy, rows, Cc, eps, momentum, moving_mean, moving_var, relu):
    """bn_finalize + bn_apply_fwd; one launch when the statistics vectors are 16-byte aligned (a slice of a grouped message may not be)"""
    aligned = all(t is None or t.data_ptr() % 16 == 0 for t in (packed, moving_mean, moving_var))
    if not aligned:
        mean, rstd = bn_finalize(packed, Cc, eps, momentum, moving_mean, moving_var)
        bn_apply_fwd(x2d, ldx, mean, rstd, gamma, beta, y2d, ldy, rows, Cc, relu)
        return mean, rstd
    mean = torch.empty(Cc, dtype=torch.float32, device=packed.device)
    rstd = torch.empty(Cc, dtype=torch.float32, device=packed.device)
    _hip.call("iseg_bn_apply_fwd_packed", ptr(x2d), ldx, ptr(packed), eps, momentum, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
              ptr(moving_mean), ptr(moving_var), ptr(y2d), ldy, rows, Cc, int(relu), dt(x2d), stream())
    return mean, rstd


def bn_apply_fwd(x2d, ldx, mean, rstd, gamma, beta, y2d, ldy, rows, Cc, relu):
    _hip.call("iseg_bn_apply_fwd", ptr(x2d), ldx, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(y2d), ldy, rows, Cc, int(relu),
              dt(x2d), stream())
    return y2d


def bn_bwd_reduce(dy2d, lddy, x2d, ldx, y2d, ldy, mean, rstd, rows, Cc, relu, out=None):
    sums = out if out is not None else torch.empty(2 * Cc, dtype=torch.float32, device=x2d.device)
    need = _hip.lib().iseg_bn_workspace_bytes(rows, Cc)
    ws, wsb = workspace(need, x2d.device)
    _hip.call("iseg_bn_bwd_reduce", ptr(dy2d), lddy, ptr(x2d), ldx, ptr(y2d), ldy, ptr(mean), ptr(rstd), ptr(sums), rows, Cc,
              int(relu), dt(x2d), ptr(ws), wsb, stream())
    return sums


def bn_bwd_apply(dy2d, lddy, x2d, ldx, y2d, ldy, mean, rstd, gamma, sums, inv_n, dx2d, lddx, rows, Cc, relu, dgamma=None, dbeta=None):
    """dgamma / dbeta (+)= the packed sums, booked by the same launch -- only valid while `sums` are this replica's own"""
    if dgamma is None and dbeta is None:
        _hip.call("iseg_bn_bwd_apply", ptr(dy2d), lddy, ptr(x2d), ldx, ptr(y2d), ldy, ptr(mean), ptr(rstd), ptr(gamma), ptr(sums), inv_n,
                  ptr(dx2d), lddx, rows, Cc, int(relu), dt(x2d), stream())
    else:
        _hip.call("iseg_bn_bwd_apply_acc", ptr(dy2d), lddy, ptr(x2d), ldx, ptr(y2d), ldy, ptr(mean), ptr(rstd), ptr(gamma), ptr(sums), inv_n,
                  ptr(dx2d), lddx, ptr(dgamma), ptr(dbeta), rows, Cc, int(relu), dt(x2d), stream())
    return dx2d


def bn_bwd_reduce_remask(dy2d, lddy, x2d, ldx, mean, rstd, gamma, beta, rows, Cc, out=None):
    """bn_bwd_reduce with fused ReLU whose mask is re-derived from x (the forward kept no output)"""
    sums = out if out is not None else torch.empty(2 * Cc, dtype=torch.float32, device=x2d.device)
    need = _hip.lib().iseg_bn_workspace_bytes(rows, Cc)
    ws, wsb = workspace(need, x2d.device)
    _hip.call("iseg_bn_bwd_reduce_remask", ptr(dy2d), lddy, ptr(x2d), ldx, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(sums), rows, Cc,
              dt(x2d), ptr(ws), wsb, stream())
    return sums


def bn_bwd_apply_remask(dy2d, lddy, x2d, ldx, mean, rstd, gamma, beta, sums, inv_n, dx2d, lddx, rows, Cc, dgamma=None, dbeta=None):
    _hip.call("iseg_bn_bwd_apply_remask", ptr(dy2d), lddy, ptr(x2d), ldx, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(sums), inv_n,
              ptr(dx2d), lddx, ptr(dgamma), ptr(dbeta), rows, Cc, dt(x2d), stream())
    return dx2d


def bn_relu_upsample_add(z, mean, rstd, gamma, beta, x):
    """relu(bn(z)) + bilinear_up(x) in one pass: z [N, Ho, Wo, C], x [N, Hi, Wi, C] (iseg_bn_relu_upsample_add)"""
    _require_cuda(z, x)
    N, Ho, Wo, Cc = z.shape
    _, Hi, Wi, _ = x.shape
    out = torch.empty_like(z)
    _hip.call("iseg_bn_relu_upsample_add", ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(x), ptr(out), N, Hi, Wi, Ho, Wo, Cc, dt(z),
              stream())
    return out


def rsqrt_eps(var, eps):
    out = torch.empty_like(var)
    _hip.call("iseg_rsqrt_eps", ptr(var), eps, ptr(out), var.numel(), stream())
    return out


# ---------------------------------------------------------------------------------------------------------
# depthwise conv
# ---------------------------------------------------------------------------------------------------------
def dwconv2d(x, w, bias, K, dil, pad_t, pad_l, *, flip=False, add=None):
    """x [N,H,W,C]; w [K*K, C] fp32 (Keras [K,K,C,1]); stride 1."""
    _require_cuda(x)
    N, H, W, Cc = x.shape
    y = torch.empty_like(x)
    timer = KERNEL_TIMER[0]
    if timer is not None:      # (bench.py's roofline_memory entry: the depthwise family is the largest non-GEMM group of the flagship step)
        timer.begin(("dwconv", int(N), int(H), int(W), int(Cc), int(K), int(dil), bool(flip), add is not None, x.dtype))
    _hip.call("iseg_dwconv2d_fwd", ptr(x), ptr(w), ptr(bias), ptr(add), ptr(y), N, H, W, Cc, K, dil, pad_t, pad_l, int(flip), dt(x),
              stream())
    if timer is not None:
        timer.end()
    return y


def dwconv2d7_mfma(x, w, bias, pad_t=3, pad_l=3, *, flip=False, add=None):
    """the 7 x 7 depthwise convolution on the matrix cores (csrc/dwconv_mfma.hip), named explicitly; dwconv2d takes this route by itself for large planes"""
    _require_cuda(x)
    N, H, W, Cc = x.shape
    y = torch.empty_like(x)
    _hip.call("iseg_dwconv2d7_mfma", ptr(x), ptr(w), ptr(bias), ptr(add), ptr(y), N, H, W, Cc, pad_t, pad_l, int(flip), stream())
    return y


def dwconv2d_bwd_weight(x, dy, dw, db, K, dil, pad_t, pad_l, accumulate=True):
    N, H, W, Cc = x.shape
    need = _hip.lib().iseg_dwconv2d_bwd_weight_workspace_bytes(N, H, W, Cc, K)
    ws, wsb = workspace(need, x.device)
    _hip.call("iseg_dwconv2d_bwd_weight", ptr(x), ptr(dy), ptr(dw), ptr(db), int(accumulate), N, H, W, Cc, K, dil, pad_t, pad_l,
              dt(x), ptr(ws), wsb, stream())


def dwconv2d_strided(x, w, bias, K, stride, dil):
    """DepthwiseConv2D(strides=stride, padding='same') at the strided positions only; x [N,H,W,C], w [K*K, C] fp32"""
    _require_cuda(x, w)
    N, H, W, Cc = x.shape
    Ho, pt = same_pad(H, K, stride, dil)
    Wo, pl = same_pad(W, K, stride, dil)
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    _hip.call("iseg_dwconv2d_strided_fwd", ptr(x), ptr(w), ptr(bias), ptr(y), N, H, W, Cc, K, stride, dil, pt, pl, Ho, Wo, dt(x), stream())
    return y


def dwconv2d_strided_bwd_data(dy, w, K, stride, dil, H, W):
    _require_cuda(dy, w)
    N, Ho, Wo, Cc = dy.shape
    _, pt = same_pad(H, K, stride, dil)
    _, pl = same_pad(W, K, stride, dil)
    dx = torch.empty((N, H, W, Cc), dtype=dy.dtype, device=dy.device)
    _hip.call("iseg_dwconv2d_strided_bwd_data", ptr(dy), ptr(w), ptr(dx), N, H, W, Cc, K, stride, dil, pt, pl, Ho, Wo, dt(dy), stream())
    return dx


def dwconv2d_strided_bwd_weight(x, dy, dw, db, K, stride, dil, accumulate=True):
    _require_cuda(x, dy, dw)
    N, H, W, Cc = x.shape
    _, Ho, Wo, _ = dy.shape
    _, pt = same_pad(H, K, stride, dil)
    _, pl = same_pad(W, K, stride, dil)
    need = _hip.lib().iseg_dwconv2d_strided_bwd_weight_workspace_bytes(N, Ho, Wo, Cc, K)
    ws, wsb = workspace(need, x.device)
    _hip.call("iseg_dwconv2d_strided_bwd_weight", ptr(x), ptr(dy), ptr(dw), ptr(db), int(accumulate), N, H, W, Cc, K, stride, dil, pt, pl, Ho, Wo,
              dt(x), ptr(ws), wsb, stream())


# ---------------------------------------------------------------------------------------------------------
# layout / elementwise
# ---------------------------------------------------------------------------------------------------------
def cast(src, dst_dtype, out=None):
    if out is None:
        out = torch.empty(src.shape, dtype=dst_dtype, device=src.device)
    _hip.call("iseg_cast", ptr(src), dt(src), ptr(out), dt(out), src.numel(), stream())
    return out


def scale_cols_cast(src, colscale, dst_dtype):
    rows, cols = src.shape
    out = torch.empty((rows, cols), dtype=dst_dtype, device=src.device)
    _hip.call("iseg_scale_cols_cast", ptr(src), ptr(colscale), ptr(out), rows, cols, dt(out), stream())
    return out


def same_pad(in_size, k, s, d):
    """TF padding="same": returns (out_size, pad_before)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - in_size, 0)
    return out, total // 2


def im2col(x, KH, KW, sh, sw, dh, dw, pt, pl, Ho, Wo, out_dtype, ldc=None):
    N, H, W, Cc = x.shape
    Kd = KH * KW * Cc
    if ldc is None:
        ldc = (Kd + 7) // 8 * 8
    col = torch.empty((N * Ho * Wo, ldc), dtype=out_dtype, device=x.device)
    _hip.call("iseg_im2col", ptr(x), dt(x), ptr(col), dt(col), N, H, W, Cc, KH, KW, sh, sw, dh, dw, pt, pl, Ho, Wo, ldc, stream())
    return col


def col2im(dcol, N, H, W, Cc, KH, KW, sh, sw, dh, dw, pt, pl, Ho, Wo):
    dx = torch.empty((N, H, W, Cc), dtype=dcol.dtype, device=dcol.device)
    _hip.call("iseg_col2im", ptr(dcol), ptr(dx), N, H, W, Cc, KH, KW, sh, sw, dh, dw, pt, pl, Ho, Wo, dcol.stride(0), dt(dcol),
              stream())
    return dx


def conv_geom(N, H, W, Cin, Cout, KH, KW, sh, sw, dh, dw, pt, pl, Ho, Wo, groups=1):
    return _hip.ConvGeom(N, H, W, Cin, Cout, KH, KW, sh, sw, dh, dw, pt, pl, Ho, Wo, groups)


def conv2d_igemm_supported(geom, dtype):
    """implicit-GEMM convolution (csrc/conv_igemm.hip): bf16 storage, channels per group multiples of 8"""
    return dtype in _DT and bool(_hip.lib().iseg_conv2d_igemm_supported(C.byref(geom), _DT[dtype]))


def _conv_ws(geom, which, device):
    return workspace(_hip.lib().iseg_conv2d_igemm_workspace_bytes(C.byref(geom), which), device)


def conv2d_igemm_fwd(x, w, bias, geom):
    """x [N,H,W,Cin] bf16, w [KH,KW,Cin/groups,Cout] bf16 -> y [N,Ho,Wo,Cout]; the patch matrix is gathered on the fly"""
    _require_cuda(x, w)
    y = torch.empty((geom.N, geom.Ho, geom.Wo, geom.Cout), dtype=x.dtype, device=x.device)
    ws, wsb = _conv_ws(geom, 0, x.device)
    _hip.call("iseg_conv2d_igemm_fwd", ptr(x), ptr(w), ptr(bias), ptr(y), C.byref(geom), dt(x), ptr(ws), wsb, stream())
    return y


def conv2d_igemm_fwd_kt_supported(geom, dtype):
    return dtype in _DT and bool(_hip.lib().iseg_conv2d_igemm_fwd_kt_supported(C.byref(geom), _DT[dtype]))


def conv2d_igemm_fwd_kt(x, wt, bias, geom):
    """the forward pass on the LDS-DMA pipeline: wt = the K-contiguous kernel copy [Cout, KH*KW*Cin] (nn.wt)"""
    _require_cuda(x, wt)
    y = torch.empty((geom.N, geom.Ho, geom.Wo, geom.Cout), dtype=x.dtype, device=x.device)
    ws, wsb = _conv_ws(geom, 0, x.device)
    _hip.call("iseg_conv2d_igemm_fwd_kt", ptr(x), ptr(wt), ptr(bias), ptr(y), C.byref(geom), dt(x), ptr(ws), wsb, stream())
    return y


def conv2d_igemm_bwd_data(dy, w, geom):
    _require_cuda(dy, w)
    dx = torch.empty((geom.N, geom.H, geom.W, geom.Cin), dtype=dy.dtype, device=dy.device)
    ws, wsb = _conv_ws(geom, 1, dy.device)
    _hip.call("iseg_conv2d_igemm_bwd_data", ptr(dy), ptr(w), ptr(dx), C.byref(geom), dt(dy), ptr(ws), wsb, stream())
    return dx


def conv2d_igemm_bwd_weight(x, dy, dw, geom, accumulate=True):
    """dw [KH,KW,Cin/groups,Cout] fp32 (+)= per-tap x^T dy"""
    _require_cuda(x, dy, dw)
    ws, wsb = _conv_ws(geom, 2, x.device)
    _hip.call("iseg_conv2d_igemm_bwd_weight", ptr(x), ptr(dy), ptr(dw), int(accumulate), C.byref(geom), dt(x), ptr(ws), wsb, stream())
    return dw


def colsum(x, ldx, batch_stride, batch, rows, Cc, out, scale=1.0, accumulate=False):
    if Cc > 8192 and batch == 1 and scale == 1.0:
        # few rows, very wide (token-axis sums, score gradients): the LDS-staged kernel would need C floats of LDS per workgroup
        L = _hip.lib()
        ws, wsb = workspace(L.iseg_colsum_wide_workspace_bytes(rows, Cc), x.device)
        _hip.check(L.iseg_colsum_wide(ptr(x), ldx, rows, Cc, ptr(out), int(accumulate), dt(x), ptr(ws), wsb, stream()), "iseg_colsum_wide")
        return out
    need = _hip.lib().iseg_colsum_workspace_bytes(batch, rows, Cc)
    ws, wsb = workspace(need, x.device)
    _hip.call("iseg_colsum", ptr(x), ldx, batch_stride, batch, rows, Cc, ptr(out), scale, int(accumulate), dt(x), ptr(ws), wsb,
              stream())
    return out


def accumulate_pair(src, n, dst0, dst1):
    """dst0 += src[:n]; dst1 += src[n:2n]   (fp32; None destinations are skipped)"""
    if dst0 is None and dst1 is None:
        return
    _hip.call("iseg_accumulate_pair", ptr(src), int(n), ptr(dst0), ptr(dst1), stream())


def broadcast_rows(v, y, ldy, batch_stride, batch, rows, Cc, scale=1.0, accumulate=False):
    _hip.call("iseg_broadcast_rows", ptr(v), dt(v), ptr(y), ldy, batch_stride, batch, rows, Cc, scale, int(accumulate), dt(y),
              stream())
    return y


def axpby(a, b, alpha=1.0, beta=1.0, out=None):
    if out is None:
        out = torch.empty_like(a)
    _hip.call("iseg_axpby", ptr(a), ptr(b), ptr(out), alpha, beta, a.numel(), dt(a), stream())
    return out


def rowscale(x2d, s, rows_per_group):
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    _hip.call("iseg_rowscale", ptr(x2d), ptr(s), ptr(y), rows, Cc, rows_per_group, dt(x2d), stream())
    return y


# Device-resident addend of every dropout / drop-path seed.  None outside graph capture; iseg_amd.graphs.GraphedTrainStep points it at a one-word
# device buffer while it captures a training step and moves the word on before every replay, so the replayed launches (frozen
# arguments) draw the numbers the eager step would have drawn.
_SEED_OFFSET = [None]


def set_seed_offset(t):
    _SEED_OFFSET[0] = t


def dropout(x, rate, seed):
    y = torch.empty_like(x)
    _hip.call("iseg_dropout", ptr(x), ptr(y), x.numel(), rate, seed, ptr(_SEED_OFFSET[0]), dt(x), stream())
    return y


def drop_path_mask(n, keep_prob, seed, device):
    s = torch.empty(n, dtype=torch.float32, device=device)
    _hip.call("iseg_drop_path_mask", ptr(s), n, keep_prob, seed, ptr(_SEED_OFFSET[0]), stream())
    return s


def drop_path_masks(keep_probs_dev, n, seed):
    """[P, n] per-sample drop-path factors, row p with keep probability keep_probs_dev[p]: one launch for a whole step"""
    P = keep_probs_dev.numel()
    s = torch.empty((P, n), dtype=torch.float32, device=keep_probs_dev.device)
    _hip.call("iseg_drop_path_masks", ptr(s), ptr(keep_probs_dev), P, n, seed, ptr(_SEED_OFFSET[0]), stream())
    return s


def fill_f32(t, value):
    _hip.call("iseg_fill_f32", ptr(t), value, t.numel(), stream())
    return t


def act_fwd(x, act):
    y = torch.empty_like(x)
    _hip.call("iseg_act_fwd", ptr(x), ptr(y), x.numel(), act, dt(x), stream())
    return y


def act_bwd(dy, aux, act):
    dx = torch.empty_like(dy)
    _hip.call("iseg_act_bwd", ptr(dy), ptr(aux), ptr(dx), dy.numel(), act, dt(dy), stream())
    return dx


def dcnv2_sample_fwd(x, offset):
    """x [N,H,W,C], offset [N,H,W,27] -> col [N,H,W,9,C] (csrc/dcnv3.hip dcnv2_sample_*; layers/dcn_v2.py:114-229)"""
    _require_cuda(x, offset)
    N, H, W, Cc = x.shape
    col = torch.empty((N, H, W, 9, Cc), dtype=x.dtype, device=x.device)
    _hip.call("iseg_dcnv2_sample_fwd", ptr(x), ptr(offset), ptr(col), N, H, W, Cc, dt(x), stream())
    return col


def dcnv2_sample_bwd(x, offset, dcol):
    N, H, W, Cc = x.shape
    dx = torch.empty((N, H, W, Cc), dtype=torch.float32, device=x.device)
    doff = torch.empty_like(offset)
    ws, wsb = workspace(_hip.lib().iseg_dcnv2_sample_bwd_workspace_bytes(N, H, W, Cc), x.device)
    _hip.call("iseg_dcnv2_sample_bwd", ptr(x), ptr(offset), ptr(dcol), ptr(dx), ptr(doff), N, H, W, Cc, dt(x), ptr(ws), wsb, stream())
    return dx, doff


def qkv_rope(qkv, q_bias, v_bias, emb, tokens, prefix, C, head_dim, inverse=False, out=None):
    """packed attention rows [B * tokens, 3 C] (csrc/eva.hip): q / v bias, rotary embedding on q and k of the tokens >= prefix; out=None: in place"""
    _require_cuda(qkv)
    out = qkv if out is None else out
    _hip.call("iseg_qkv_rope", ptr(qkv), ptr(out), ptr(q_bias), ptr(v_bias), ptr(emb), qkv.numel() // (3 * C), tokens, prefix, C, head_dim, int(inverse),
              dt(qkv), stream())
    return out


def glu_fwd(gate, x, act):
    """act(gate) * x; gate / x: [rows, cols] views with unit column stride (row strides free)"""
    rows, cols = gate.shape
    out = torch.empty((rows, cols), dtype=gate.dtype, device=gate.device)
    _hip.call("iseg_glu_fwd", ptr(gate), gate.stride(0), ptr(x), x.stride(0), ptr(out), cols, rows, cols, act, dt(gate), stream())
    return out


def glu_bwd(dout, gate, x, dgate, dx, act):
    rows, cols = gate.shape
    _hip.call("iseg_glu_bwd", ptr(dout), dout.stride(0), ptr(gate), gate.stride(0), ptr(x), x.stride(0), ptr(dgate), dgate.stride(0), ptr(dx),
              dx.stride(0), rows, cols, act, dt(gate), stream())


def copy2d(src, ld_src, dst, ld_dst, rows, cols):
    _hip.call("iseg_copy2d", ptr(src), ld_src, ptr(dst), ld_dst, rows, cols, dt(src), stream())
    return dst


def add2d(src, ld_src, dst, ld_dst, rows, cols):
    _hip.call("iseg_add2d_f32", ptr(src), ld_src, ptr(dst), ld_dst, rows, cols, stream())


def scale_rows(x2d, s):
    y = torch.empty_like(x2d)
    _hip.call("iseg_scale_rows_f32", ptr(x2d), ptr(s), ptr(y), x2d.shape[0], x2d.shape[1], stream())
    return y


def layerscale_grads(Z, W2, b2, gamma, S, dW2, dgamma, db2, accumulate=True):
    Kd, Nd = Z.shape
    need = _hip.lib().iseg_layerscale_grads_workspace_bytes(Kd, Nd)
    ws, wsb = workspace(need, Z.device)
    _hip.call("iseg_layerscale_grads", ptr(Z), ptr(W2), ptr(b2), ptr(gamma), ptr(S), ptr(dW2), ptr(dgamma), ptr(db2), Kd, Nd,
              int(accumulate), ptr(ws), wsb, stream())


def layerscale_grads_slabs(slabs, nslabs, W2, b2, gamma, dW2, dgamma, db2, accumulate=True, extra=None, srow=None):
    """layerscale_grads from the unreduced split-K slabs of Z = g^T dout (dense_wgrad_slabs): Z and S = colsum(dout) are summed on load.
    extra = dense_wgrad_pair(..., defer_second=True)[2]: the other product's slabs are summed into their gradients by the same launch"""
    Kd, Nd = W2.shape
    need = _hip.lib().iseg_layerscale_grads_workspace_bytes(Kd, Nd)
    ws, wsb = workspace(need, slabs.device)
    if srow is not None:
        # srow = (dout2d bf16 [M, C], rowscale or None, rows_per_group): the slabs carry no ones-row; S = colsum(rowscale * dout) is formed in this launch
        dout2d, rs, rpg = srow
        part2, p2, n2, out0, out1, n0, acc2 = extra if extra is not None else (None, 0, 0, None, None, 0, False)
        _hip.call("iseg_layerscale_grads_slabs_srow", ptr(slabs), int(nslabs), ptr(W2), ptr(b2), ptr(gamma), ptr(dW2), ptr(dgamma), ptr(db2), Kd, Nd,
                  int(accumulate), ptr(ws), wsb, ptr(part2), int(p2), int(n2), ptr(out0), ptr(out1), int(n0), int(acc2), ptr(dout2d),
                  dout2d.stride(0), dout2d.shape[0], ptr(rs), int(rpg if rs is not None else 1), stream())
        return
    if extra is not None:
        part2, p2, n2, out0, out1, n0, acc2 = extra
        _hip.call("iseg_layerscale_grads_slabs_reduce", ptr(slabs), int(nslabs), ptr(W2), ptr(b2), ptr(gamma), ptr(dW2), ptr(dgamma), ptr(db2), Kd,
                  Nd, int(accumulate), ptr(ws), wsb, ptr(part2), int(p2), int(n2), ptr(out0), ptr(out1), int(n0), int(acc2), stream())
        return
    _hip.call("iseg_layerscale_grads_slabs", ptr(slabs), int(nslabs), ptr(W2), ptr(b2), ptr(gamma), ptr(dW2), ptr(dgamma), ptr(db2), Kd, Nd,
              int(accumulate), ptr(ws), wsb, stream())


# ---------------------------------------------------------------------------------------------------------
# input pipeline (csrc/augment.hip)
# ---------------------------------------------------------------------------------------------------------
def augment_params_ints():
    return int(_hip.lib().iseg_augment_params_ints())


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def augment_params_floats():
    return int(_hip.lib().iseg_augment_params_floats())


def augment_channel_means(images, params, fparams):
    """fill slots 2..4 of the photometric table with the channel means of each sample's scaled (+ brightness) image (tf.image.adjust_contrast)"""
    _require_cuda(images, params, fparams)
    B, Hs, Ws, _ = images.shape
    need = _hip.lib().iseg_augment_means_workspace_bytes(B)
    ws, wsb = workspace(need, images.device)
    _hip.call("iseg_augment_channel_means", ptr(images.contiguous()), 0 if images.dtype == torch.float32 else 2, ptr(params), ptr(fparams), B, Hs, Ws,
              ptr(ws), wsb, stream())


def augment_crop_batch(images, labels, params, mean_pixel, norm_scale, norm_shift, ignore_label, crop_h, crop_w, seed, fparams=None):
    """scale -> [brightness / contrast / saturation / hue] -> pad -> crop -> flip -> erase -> [eval noise] -> normalise as one gather
    (data_process/pipeline.py:85-170); params: int32 [B, n], fparams: float32 [B, m] or None, both on the device"""
    _require_cuda(images, params)
    if images.dtype not in (torch.float32, torch.uint8):
        raise TypeError("augment_crop_batch takes float32 or uint8 images")
    B, Hs, Ws, _ = images.shape
    images = images.contiguous()
    out = torch.empty((B, crop_h, crop_w, 3), dtype=torch.float32, device=images.device)
    lab = out_lab = None
    if labels is not None:
        lab = labels.contiguous()
        if lab.dtype != torch.int32:
            lab = lab.to(torch.int32)
        out_lab = torch.empty((B, crop_h, crop_w), dtype=torch.int32, device=images.device)
    _hip.call("iseg_augment_crop_batch", ptr(images), 0 if images.dtype == torch.float32 else 2, ptr(lab), ptr(params.contiguous()),
              ptr(fparams), _f3(mean_pixel), _f3(norm_scale), _f3(norm_shift), int(ignore_label), ptr(out), ptr(out_lab), B, Hs, Ws,
              int(crop_h), int(crop_w), int(seed), stream())
    return out, out_lab


def normalize_image(x, norm_scale, norm_shift):
    _require_cuda(x)
    y = torch.empty_like(x)
    _hip.call("iseg_normalize_image", ptr(x), ptr(y), x.numel() // 3, _f3(norm_scale), _f3(norm_shift), stream())
    return y


# ---------------------------------------------------------------------------------------------------------
# resize / loss / metric / optimizer
# ---------------------------------------------------------------------------------------------------------
def resize_bilinear(x, Ho, Wo, out_dtype=None, align_corners=False):
    """align_corners: tf.compat.v1.image.resize(..., align_corners=True) coordinates instead of TF2's half-pixel centres"""
    _require_cuda(x)
    N, Hi, Wi, Cc = x.shape
    y = torch.empty((N, Ho, Wo, Cc), dtype=out_dtype or x.dtype, device=x.device)
    _hip.call("iseg_resize_bilinear_ac_fwd" if align_corners else "iseg_resize_bilinear_fwd", ptr(x), dt(x), ptr(y), dt(y), N, Hi, Wi, Ho, Wo,
              Cc, stream())
    return y


def resize_bilinear_bwd(dy, Hi, Wi, dx_dtype, dx_add=None, align_corners=False):
    N, Ho, Wo, Cc = dy.shape
    dx = torch.empty((N, Hi, Wi, Cc), dtype=dx_dtype, device=dy.device)
    need = _hip.lib().iseg_resize_bilinear_bwd_workspace_bytes(N, Hi, Wi, Ho, Wo, Cc)
    ws, wsb = workspace(need, dy.device)
    _hip.call("iseg_resize_bilinear_ac_bwd" if align_corners else "iseg_resize_bilinear_bwd", ptr(dy), dt(dy), ptr(dx), dt(dx), ptr(dx_add), N,
              Hi, Wi, Ho, Wo, Cc, ptr(ws), wsb, stream())
    return dx


def resize_nearest_i32(x, Ho, Wo):
    N, Hi, Wi, Cc = x.shape
    y = torch.empty((N, Ho, Wo, Cc), dtype=torch.int32, device=x.device)
    _hip.call("iseg_resize_nearest_i32", ptr(x), ptr(y), N, Hi, Wi, Ho, Wo, Cc, stream())
    return y


def scale_dev(x, s_dev):
    y = torch.empty_like(x)
    _hip.call("iseg_scale_dev", ptr(x), ptr(s_dev), ptr(y), x.numel(), dt(x), stream())
    return y


def softmax_ce_ignore(logits2d, labels1d, ignore_label, *, class_w=None, want_px=True, want_sum=False, sum_scale=1.0,
                      want_grad=False, grad_scale=1.0, grad_px=None, focal=None, cm=None):
    """focal = (alpha, gamma) selects keras' CategoricalFocalCrossentropy instead of the plain cross-entropy; cm (int64 [C*C]) also
    accumulates the confusion matrix of argmax(logits) in the same pass (plain cross-entropy only)"""
    _require_cuda(logits2d, labels1d)
    P, Cc = logits2d.shape
    dev = logits2d.device
    loss_px = torch.empty(P, dtype=torch.float32, device=dev) if want_px else None
    loss_sum = torch.empty(1, dtype=torch.float32, device=dev) if want_sum else None
    dlogits = torch.empty_like(logits2d) if want_grad else None
    need = _hip.lib().iseg_softmax_ce_workspace_bytes(P, Cc) if want_sum else 0
    ws, wsb = workspace(need, dev)
    if cm is not None:
        if focal is not None:
            raise ValueError("softmax_ce_ignore: cm rides the plain cross-entropy kernel only")
        _hip.call("iseg_softmax_ce_confusion", ptr(logits2d), ptr(labels1d), ptr(class_w), P, Cc, ignore_label, ptr(loss_px), ptr(loss_sum),
                  sum_scale, ptr(dlogits), grad_scale, ptr(grad_px), ptr(cm), ptr(ws), wsb, stream())
    elif focal is not None:
        _hip.call("iseg_softmax_focal_ce_ignore", ptr(logits2d), ptr(labels1d), ptr(class_w), P, Cc, ignore_label, float(focal[0]),
                  float(focal[1]), ptr(loss_px), ptr(loss_sum), sum_scale, ptr(dlogits), grad_scale, ptr(grad_px), ptr(ws), wsb, stream())
    else:
        _hip.call("iseg_softmax_ce_ignore", ptr(logits2d), ptr(labels1d), ptr(class_w), P, Cc, ignore_label, ptr(loss_px), ptr(loss_sum),
                  sum_scale, ptr(dlogits), grad_scale, ptr(grad_px), ptr(ws), wsb, stream())
    return loss_px, loss_sum, dlogits


def upsample_ce_supported(Hi, Wi, Ho, Wo, Cc):
    return bool(_hip.lib().iseg_upsample_ce_supported(int(Hi), int(Wi), int(Ho), int(Wo), int(Cc)))


def upsample_ce(z, labels, Ho, Wo, ignore_label, *, class_w=None, sum_scale=1.0, want_grad=True, grad_scale=1.0, cm=None):
    """fused logits tail (csrc/loss.hip): bilinear upsample of z [N,Hi,Wi,C] to [Ho,Wo] + ignore-label CE sum + d(sum)/dz + confusion
    matrix, without the full-resolution logits.  returns (loss_sum [1] fp32, dz like z or None)"""
    _require_cuda(z, labels)
    N, Hi, Wi, Cc = z.shape
    loss_sum = torch.empty(1, dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z) if want_grad else None
    ws, wsb = workspace(_hip.lib().iseg_upsample_ce_workspace_bytes(N, Hi, Wi, Ho, Wo, Cc), z.device)
    _hip.call("iseg_upsample_ce", ptr(z), dt(z), ptr(labels), ptr(class_w), N, Hi, Wi, Ho, Wo, Cc, ignore_label, ptr(loss_sum), sum_scale,
              ptr(dz), grad_scale, ptr(cm), ptr(ws), wsb, stream())
    return loss_sum, dz


def argmax_confusion(logits2d, labels1d, ignore_label, cm=None, want_pred=False):
    P, Cc = logits2d.shape
    pred = torch.empty(P, dtype=torch.int32, device=logits2d.device) if want_pred else None
    _hip.call("iseg_argmax_confusion", ptr(logits2d), ptr(labels1d), P, Cc, ignore_label, ptr(pred), ptr(cm), stream())
    return pred


# ---------------------------------------------------------------------------------------------------------
# replace_nan_or_inf, GroupNorm, RMSNorm, pooling (csrc/misc.hip)
# ---------------------------------------------------------------------------------------------------------
def replace_nan_or_inf(x, nan_value=0.0):
    _require_cuda(x)
    y = torch.empty_like(x)
    ws, wsb = workspace(8, x.device)
    _hip.check(_hip.lib().iseg_replace_nan_or_inf(ptr(x), ptr(y), x.numel(), float(nan_value), dt(x), ptr(ws), wsb, stream()),
               "iseg_replace_nan_or_inf")
    return y


def replace_nan_or_inf_bwd(x, dy):
    _require_cuda(x, dy)
    dx = torch.empty_like(dy)
    _hip.check(_hip.lib().iseg_replace_nan_or_inf_bwd(ptr(x), ptr(dy), ptr(dx), x.numel(), dt(x), stream()), "iseg_replace_nan_or_inf_bwd")
    return dx


def groupnorm_fwd(x3d, gamma, beta, groups, eps):
    """x3d [N, HW, C] contiguous -> y, mean [N*G], rstd [N*G]"""
    _require_cuda(x3d)
    N, HW, Cc = x3d.shape
    y = torch.empty_like(x3d)
    mean = torch.empty(N * groups, dtype=torch.float32, device=x3d.device)
    rstd = torch.empty_like(mean)
    _hip.check(_hip.lib().iseg_groupnorm_fwd(ptr(x3d), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), N, HW, Cc, groups, float(eps),
                                            dt(x3d), stream()), "iseg_groupnorm_fwd")
    return y, mean, rstd


def groupnorm_bwd(dy3d, x3d, gamma, mean, rstd, groups, dgamma, dbeta, accumulate=True):
    _require_cuda(dy3d, x3d)
    N, HW, Cc = x3d.shape
    dx = torch.empty_like(x3d)
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_groupnorm_bwd_workspace_bytes(N, Cc), x3d.device)
    _hip.check(L.iseg_groupnorm_bwd(ptr(dy3d), ptr(x3d), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), int(accumulate),
                                    N, HW, Cc, groups, dt(x3d), ptr(ws), wsb, stream()), "iseg_groupnorm_bwd")
    return dx


def rmsnorm_fwd(x2d, scale, eps):
    _require_cuda(x2d)
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    _hip.check(_hip.lib().iseg_rmsnorm_fwd(ptr(x2d), ptr(scale), ptr(y), ptr(rstd), rows, Cc, float(eps), dt(x2d), stream()), "iseg_rmsnorm_fwd")
    return y, rstd


def rmsnorm_bwd(dy2d, x2d, scale, rstd, dscale, accumulate=True):
    _require_cuda(dy2d, x2d)
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_rmsnorm_bwd_workspace_bytes(rows, Cc), x2d.device)
    _hip.check(L.iseg_rmsnorm_bwd(ptr(dy2d), ptr(x2d), ptr(scale), ptr(rstd), ptr(dx), ptr(dscale), int(accumulate), rows, Cc, dt(x2d),
                                  ptr(ws), wsb, stream()), "iseg_rmsnorm_bwd")
    return dx


def grn_fwd(x3d, gamma, beta, eps):
    """Global Response Normalization (backbones/convnext_v2.py:45-60) on [N, HW, C]: returns y and the per-(sample, channel) nx, gx"""
    _require_cuda(x3d, gamma, beta)
    N, HW, Cc = x3d.shape
    y = torch.empty_like(x3d)
    nx = torch.empty(N, Cc, dtype=torch.float32, device=x3d.device)
    gx = torch.empty_like(nx)
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_grn_workspace_bytes(N, HW, Cc), x3d.device)
    _hip.check(L.iseg_grn_fwd(ptr(x3d), ptr(gamma), ptr(beta), ptr(y), ptr(nx), ptr(gx), N, HW, Cc, float(eps), dt(x3d), ptr(ws), wsb,
                              stream()), "iseg_grn_fwd")
    return y, nx, gx


def grn_bwd(dy3d, x3d, gamma, nx, gx, dgamma, dbeta, eps, accumulate=True, mul=None):
    """dx [* mul] and (+)= dgamma, dbeta; `mul`: saved activation derivative folded into the data gradient (see include/iseg_hip.h)"""
    _require_cuda(dy3d, x3d, dgamma, dbeta)
    if mul is not None and (mul.dtype != x3d.dtype or mul.numel() != x3d.numel() or not mul.is_contiguous()):
        raise ValueError("grn_bwd: mul must be a contiguous tensor shaped and typed like x")
    N, HW, Cc = x3d.shape
    dx = torch.empty_like(x3d)
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_grn_workspace_bytes(N, HW, Cc), x3d.device)
    _hip.check(L.iseg_grn_bwd(ptr(dy3d), ptr(x3d), ptr(gamma), ptr(nx), ptr(gx), ptr(mul), ptr(dx), ptr(dgamma), ptr(dbeta), int(accumulate), N, HW, Cc,
                              float(eps), dt(x3d), ptr(ws), wsb, stream()), "iseg_grn_bwd")
    return dx


def grn_stats(x3d, eps):
    """nx, gx of grn_fwd without writing the normalised tensor (the caller folds gamma*nx + 1 into the next kernel)"""
    _require_cuda(x3d)
    N, HW, Cc = x3d.shape
    nx = torch.empty(N, Cc, dtype=torch.float32, device=x3d.device)
    gx = torch.empty_like(nx)
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_grn_workspace_bytes(N, HW, Cc), x3d.device)
    _hip.check(L.iseg_grn_fwd(ptr(x3d), None, None, None, ptr(nx), ptr(gx), N, HW, Cc, float(eps), dt(x3d), ptr(ws), wsb, stream()), "iseg_grn_fwd")
    return nx, gx


def grn_fold_weights(wt, gamma, nx):
    """[N, Cout, C4] bf16: the K-contiguous kernel copy wt [Cout, C4] times (gamma*nx_n + 1) per sample"""
    _require_cuda(wt, gamma, nx)
    Cout, C4 = wt.shape
    N = nx.shape[0]
    out = torch.empty((N, Cout, C4), dtype=wt.dtype, device=wt.device)
    _hip.call("iseg_grn_fold_weights", ptr(wt), ptr(gamma), ptr(nx), ptr(out), N, Cout, C4, stream())
    return out


def grn_fold_bias(W, beta, b):
    """b + beta @ W for the fp32 kernel W [C4, Cout]"""
    _require_cuda(W, beta)
    C4, Cout = W.shape
    out = torch.empty(Cout, dtype=torch.float32, device=W.device)
    _hip.call("iseg_grn_fold_bias", ptr(W), ptr(beta), ptr(b), ptr(out), C4, Cout, stream())
    return out


def grn_fold_wgrad(slabs, slabs_per_sample, W, gamma, beta, nx, S, dW, accumulate=True):
    """dW (+)= sum_n (gamma*nx_n + 1) (.) G_n + beta (x) S from the per-chunk products slabs [N*sps, C4, Cout]; returns dstats [N, 2*C4]"""
    _require_cuda(slabs, W, dW)
    C4, Cout = W.shape
    N = nx.shape[0]
    dstats = torch.empty((N, 2 * C4), dtype=torch.float32, device=W.device)
    _hip.call("iseg_grn_fold_wgrad", ptr(slabs), int(slabs_per_sample), ptr(W), ptr(gamma), ptr(beta), ptr(nx), ptr(S), ptr(dW), ptr(dstats),
              int(accumulate), N, C4, Cout, stream())
    return dstats


def grn_bwd_folded(dy3d, x3d, gamma, nx, gx, dstats, dgamma, dbeta, eps, accumulate=True, mul=None):
    """grn_bwd with the per-sample statistics given (dstats from grn_fold_wgrad; overwritten)"""
    _require_cuda(dy3d, x3d, dstats, dgamma, dbeta)
    N, HW, Cc = x3d.shape
    if mul is not None and (mul.dtype != x3d.dtype or mul.numel() != x3d.numel() or not mul.is_contiguous()):
        raise ValueError("grn_bwd_folded: mul must be a contiguous tensor shaped and typed like x")
    dx = torch.empty_like(x3d)
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_grn_workspace_bytes(N, HW, Cc), x3d.device)
    _hip.check(L.iseg_grn_bwd_folded(ptr(dy3d), ptr(x3d), ptr(gamma), ptr(nx), ptr(gx), ptr(mul), ptr(dstats), ptr(dx), ptr(dgamma), ptr(dbeta),
                                     int(accumulate), N, HW, Cc, float(eps), dt(x3d), ptr(ws), wsb, stream()), "iseg_grn_bwd_folded")
    return dx


POOL_MAX, POOL_AVG = 0, 1


def pool2d_fwd(x, kh, kw, sh, sw, pt, pl, Ho, Wo, mode):
    _require_cuda(x)
    N, H, W, Cc = x.shape
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    _hip.check(_hip.lib().iseg_pool2d_fwd(ptr(x), ptr(y), N, H, W, Cc, kh, kw, sh, sw, pt, pl, Ho, Wo, mode, dt(x), stream()), "iseg_pool2d_fwd")
    return y


def pool2d_bwd(x, dy, kh, kw, sh, sw, pt, pl, mode):
    _require_cuda(x, dy)
    N, H, W, Cc = x.shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    dx = torch.empty_like(x)
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_pool2d_bwd_workspace_bytes(N, Ho, Wo, Cc, mode), x.device)
    _hip.check(L.iseg_pool2d_bwd(ptr(x), ptr(dy), ptr(dx), N, H, W, Cc, kh, kw, sh, sw, pt, pl, Ho, Wo, mode, dt(x), ptr(ws), wsb,
                                 stream()), "iseg_pool2d_bwd")
    return dx


def add_relu(a, b):
    _require_cuda(a, b)
    y = torch.empty_like(a)
    _hip.check(_hip.lib().iseg_add_relu(ptr(a), ptr(b), ptr(y), a.numel(), dt(a), stream()), "iseg_add_relu")
    return y


# ---------------------------------------------------------------------------------------------------------
# attention pieces (csrc/attention.hip) around the strided-batch GEMMs
# ---------------------------------------------------------------------------------------------------------
def softmax_rows_fwd(scores, problems, Tq, cols, ld, *, bias=None, heads=1, mask=None, windows=1, clip=None, out=None):
    _require_cuda(scores)
    out = scores if out is None else out
    lo, hi = clip if clip is not None else (0.0, 0.0)
    _hip.check(_hip.lib().iseg_softmax_rows_fwd(ptr(scores), ptr(out), problems, Tq, cols, ld, ptr(bias), heads, ptr(mask), windows,
                                               float(lo), float(hi), dt(scores), stream()), "iseg_softmax_rows_fwd")
    return out


def softmax_rows_bwd(probs, dprobs, rows, cols, ld, *, clip=None, out=None):
    _require_cuda(probs, dprobs)
    out = dprobs if out is None else out
    lo, hi = clip if clip is not None else (0.0, 0.0)
    _hip.check(_hip.lib().iseg_softmax_rows_bwd(ptr(probs), ptr(dprobs), ptr(out), rows, cols, ld, float(lo), float(hi), dt(probs),
                                               stream()), "iseg_softmax_rows_bwd")
    return out


def clip_fwd(x, lo, hi):
    _require_cuda(x)
    y = torch.empty_like(x)
    _hip.check(_hip.lib().iseg_clip_fwd(ptr(x), ptr(y), x.numel(), float(lo), float(hi), dt(x), stream()), "iseg_clip_fwd")
    return y


def clip_bwd(x, dy, lo, hi):
    _require_cuda(x, dy)
    dx = torch.empty_like(dy)
    _hip.check(_hip.lib().iseg_clip_bwd(ptr(x), ptr(dy), ptr(dx), x.numel(), float(lo), float(hi), dt(x), stream()), "iseg_clip_bwd")
    return dx


def gather_rows(x2d, idx, rows_out):
    """y[r] = x2d[idx[r]] (idx int32 on device, -1 -> zero row)"""
    _require_cuda(x2d, idx)
    if idx.dtype != torch.int32 or idx.numel() != rows_out:
        raise TypeError("gather_rows: idx must be int32 with one entry per output row")
    rows_in, Cc = x2d.shape
    y = torch.empty((rows_out, Cc), dtype=x2d.dtype, device=x2d.device)
    _hip.check(_hip.lib().iseg_gather_rows(ptr(x2d), ptr(idx), ptr(y), rows_in, rows_out, Cc, dt(x2d), stream()), "iseg_gather_rows")
    return y


def relpos_bias_gather(table, index, heads, T):
    _require_cuda(table, index)
    bias = torch.empty((heads, T, T), dtype=torch.float32, device=table.device)
    _hip.check(_hip.lib().iseg_relpos_bias_gather(ptr(table), ptr(index), ptr(bias), heads, T * T, stream()), "iseg_relpos_bias_gather")
    return bias


def relpos_bias_scatter_grad(dbias, ld, index, dtable, heads, T, accumulate=True, window=0):
    """window = ws > 0 asserts that `index` is the canonical ws x ws table (backbones/swin.py:93-104): fast kernel"""
    _require_cuda(dbias, index, dtable)
    if window > 0 and window * window == T and dtable.shape[0] == (2 * window - 1) ** 2:
        _hip.check(_hip.lib().iseg_relpos_bias_scatter_grad_window(ptr(dbias), ld, ptr(dtable), window, heads, int(accumulate), stream()),
                   "iseg_relpos_bias_scatter_grad_window")
        return dtable
    _hip.check(_hip.lib().iseg_relpos_bias_scatter_grad(ptr(dbias), ld, ptr(index), ptr(dtable), dtable.shape[0], heads, T, int(accumulate),
                                                       stream()), "iseg_relpos_bias_scatter_grad")
    return dtable


# ---------------------------------------------------------------------------------------------------------
# DCNv3 core + per-channel scale gradient (csrc/dcnv3.hip)
# ---------------------------------------------------------------------------------------------------------
def dcnv3_out_hw(H, W, kh, kw, stride, dil, pad):
    return (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1, (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1


def dcnv3_fwd(x, offset, mask, G, Cg, kh, kw, stride, dil, pad, offset_scale):
    _require_cuda(x, offset, mask)
    N, H, W, Cc = x.shape
    Ho, Wo = dcnv3_out_hw(H, W, kh, kw, stride, dil, pad)
    if Cc != G * Cg or tuple(offset.shape) != (N, Ho, Wo, G * kh * kw * 2) or tuple(mask.shape) != (N, Ho, Wo, G * kh * kw):
        raise ValueError(f"dcnv3_fwd: shapes x {tuple(x.shape)} offset {tuple(offset.shape)} mask {tuple(mask.shape)} do not match "
                         f"G={G} Cg={Cg} k={kh}x{kw} -> {Ho}x{Wo}")
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    _hip.check(_hip.lib().iseg_dcnv3_fwd(ptr(x), ptr(offset), ptr(mask), ptr(y), N, H, W, G, Cg, kh, kw, stride, dil, pad,
                                        float(offset_scale), dt(x), stream()), "iseg_dcnv3_fwd")
    return y


def dcnv3_bwd(x, offset, mask, dy, G, Cg, kh, kw, stride, dil, pad, offset_scale):
    _require_cuda(x, offset, mask, dy)
    N, H, W, Cc = x.shape
    Ho, Wo = dcnv3_out_hw(H, W, kh, kw, stride, dil, pad)
    if tuple(dy.shape) != (N, Ho, Wo, Cc) or Cc != G * Cg:
        raise ValueError("dcnv3_bwd: dy shape does not match the forward geometry")
    dx = torch.empty((N, H, W, Cc), dtype=torch.float32, device=x.device)
    doff = torch.empty_like(offset)
    dmask = torch.empty_like(mask)
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_dcnv3_bwd_workspace_bytes(N, H, W, G, Cg, kh, kw, stride, dil, pad, float(offset_scale)), x.device)
    _hip.check(L.iseg_dcnv3_bwd(ptr(x), ptr(offset), ptr(mask), ptr(dy), ptr(dx), ptr(doff), ptr(dmask), N, H, W, G, Cg, kh, kw,
                                stride, dil, pad, float(offset_scale), dt(x), ptr(ws), wsb, stream()), "iseg_dcnv3_bwd")
    return dx, doff, dmask


def dcnv3_fwd_joint(x, om, G, Cg, kh, kw, stride, dil, pad, offset_scale):
    """dcnv3_fwd with the offsets and the (soft-maxed) mask as column ranges [0, 2GP) and [2GP, 3GP) of ONE matrix om [pixels, ld]"""
    _require_cuda(x, om)
    N, H, W, Cc = x.shape
    Ho, Wo = dcnv3_out_hw(H, W, kh, kw, stride, dil, pad)
    gp = G * kh * kw
    if Cc != G * Cg or om.dim() != 2 or om.shape[0] != N * Ho * Wo or om.shape[1] < 3 * gp or om.stride(1) != 1:
        raise ValueError(f"dcnv3_fwd_joint: x {tuple(x.shape)} / om {tuple(om.shape)} do not match G={G} Cg={Cg} k={kh}x{kw} -> {Ho}x{Wo}")
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    ld = om.stride(0)
    _hip.check(_hip.lib().iseg_dcnv3_fwd_ld(ptr(x), ptr(om), ptr(om[:, 2 * gp:]), ld, ld, ptr(y), N, H, W, G, Cg, kh, kw, stride, dil, pad,
                                           float(offset_scale), dt(x), stream()), "iseg_dcnv3_fwd_ld")
    return y


def dcnv3_bwd_joint(x, om, dy, G, Cg, kh, kw, stride, dil, pad, offset_scale):
    """-> (dx fp32, dom [pixels, ld]): the offset / mask gradients in om's layout; the columns behind 3GP are NOT written (dcn_mask_softmax_bwd
    clears them)"""
    _require_cuda(x, om, dy)
    N, H, W, Cc = x.shape
    Ho, Wo = dcnv3_out_hw(H, W, kh, kw, stride, dil, pad)
    if tuple(dy.shape) != (N, Ho, Wo, Cc) or Cc != G * Cg:
        raise ValueError("dcnv3_bwd_joint: dy shape does not match the forward geometry")
    gp = G * kh * kw
    if om.dim() != 2 or om.shape[0] != N * Ho * Wo or om.shape[1] < 3 * gp or om.stride(1) != 1:
        raise ValueError(f"dcnv3_bwd_joint: om {tuple(om.shape)} does not match {N * Ho * Wo} output pixels x >= {3 * gp} columns")
    dx = torch.empty((N, H, W, Cc), dtype=x.dtype, device=x.device)
    # the C ABI takes ONE pitch pair for om and dom, so dom is allocated AT om's row pitch: empty_like(om) of a column-sliced om (shape[1] < stride)
    # keeps om's shape but comes back dense, i.e. with a different pitch than the om.stride(0) that was passed for it (round-5 advisor)
    ld = om.stride(0)
    dom = torch.empty((om.shape[0], ld), dtype=om.dtype, device=om.device)[:, :om.shape[1]]
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_dcnv3_bwd_workspace_bytes(N, H, W, G, Cg, kh, kw, stride, dil, pad, float(offset_scale)), x.device)
    side = _dcn_side(L.iseg_dcnv3_bwd_side_bytes(N, H, W, G, Cg, kh, kw, stride, dil, pad, float(offset_scale)), x.device)
    _dcn_side_guard(x.device, True)
    _hip.check(L.iseg_dcnv3_bwd_ld(ptr(x), ptr(om), ptr(om[:, 2 * gp:]), ld, ld, ptr(dy), ptr(dx), dt(dx), ptr(dom), ptr(dom[:, 2 * gp:]), N, H, W, G,
                                   Cg, kh, kw, stride, dil, pad, float(offset_scale), dt(x), ptr(ws), wsb, ptr(side),
                                   side.numel() if side is not None else 0, stream()), "iseg_dcnv3_bwd_ld")
    _dcn_side_guard(x.device, False)
    return dx, dom


# The kept side buffer of the DCNv3 backward (iseg_dcnv3_bwd_ld): zero-filled once when it is (re)allocated, left all zero by every call.  One per
# device, grown to the largest geometry seen; a growth re-homes it, which a captured graph must notice (nn.buffers_generation).
_DCN_SIDE = {}


_DCN_SIDE_DIRTY = set()      # devices whose side buffer a backward call may have left non-zero


def _dcn_side_guard(device, entering):
    """Host-side dirty bit around every call that uses the side buffer (round-5 advisor): set before the launch, cleared once the call has
    returned without an error.  A call that raises in between (a launch error surfacing there, KeyboardInterrupt, a failed graph capture that is
    retried) leaves it set, and the NEXT request for the buffer zero-fills it again instead of feeding stale fixed-point partials or a set poison
    flag into every later input gradient of every DCNv3 layer."""
    key = str(device)
    if entering:
        _DCN_SIDE_DIRTY.add(key)
    else:
        _DCN_SIDE_DIRTY.discard(key)


def _dcn_side(nbytes, device):
    if nbytes == 0:
        return None
    key = str(device)
    buf = _DCN_SIDE.get(key)
    if buf is not None and key in _DCN_SIDE_DIRTY:      # a call that used it did not finish: restore the all-zero invariant
        fill_f32(buf.view(torch.float32), 0.0)
        _DCN_SIDE_DIRTY.discard(key)
    if buf is None or buf.numel() < nbytes:
        from . import nn

        buf = torch.empty((nbytes + 3) // 4 * 4, dtype=torch.uint8, device=device)
        fill_f32(buf.view(torch.float32), 0.0)
        _DCN_SIDE[key] = buf
        nn.bump_buffers_generation()
    return buf


def dcn_mask_softmax_fwd(om, G, P):
    """softmax over each (pixel, group)'s P mask logits, in place on columns [2GP, 3GP) of om [pixels, ld]"""
    _hip.call("iseg_dcn_mask_softmax_fwd", ptr(om), om.shape[0], G, P, om.stride(0), 2 * G * P, dt(om), stream())
    return om


def dcn_mask_softmax_bwd(om, dom, G, P):
    """in place on dom: the softmax backward on the mask columns, zeros in the padding columns behind 3GP"""
    _hip.call("iseg_dcn_mask_softmax_bwd", ptr(om), ptr(dom), om.shape[0], G, P, om.stride(0), 2 * G * P, dt(om), stream())
    return dom


def split_cols_accumulate(src, dst0, n0, dst1, n1):
    """dst0 [rows, n0] += src[:, :n0]; dst1 [rows, n1] += src[:, n0:n0 + n1]   (fp32; src [rows, ld]; None destinations are skipped)"""
    rows = src.shape[0] if src.dim() == 2 else 1
    ld = src.stride(0) if src.dim() == 2 else src.numel()
    _hip.call("iseg_split_cols_accumulate", ptr(src), rows, ld, ptr(dst0), int(n0), ptr(dst1), int(n1), stream())


def dcn_center_blend_fwd(x, x_proj, scale, G, Cg):
    """x (1 - s) + x_proj s, s [.., G] broadcast over each group's Cg channels"""
    _require_cuda(x, x_proj, scale)
    out = torch.empty_like(x)
    _hip.call("iseg_dcn_center_blend_fwd", ptr(x), ptr(x_proj), ptr(scale), ptr(out), x.numel() // (G * Cg), G, Cg, dt(x), stream())
    return out


def dcn_center_blend_bwd(dout, x, x_proj, scale, G, Cg):
    _require_cuda(dout, x, x_proj, scale)
    dx, dxp, ds = torch.empty_like(x), torch.empty_like(x_proj), torch.empty_like(scale)
    _hip.call("iseg_dcn_center_blend_bwd", ptr(dout), ptr(x), ptr(x_proj), ptr(scale), ptr(dx), ptr(dxp), ptr(ds), x.numel() // (G * Cg), G, Cg,
              dt(x), stream())
    return dx, dxp, ds


def mul_colsum(a2d, b2d, out, accumulate=True):
    _require_cuda(a2d, b2d, out)
    rows, Cc = a2d.shape
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_mul_colsum_workspace_bytes(rows, Cc), a2d.device)
    _hip.check(L.iseg_mul_colsum(ptr(a2d), ptr(b2d), rows, Cc, ptr(out), int(accumulate), dt(a2d), ptr(ws), wsb, stream()), "iseg_mul_colsum")
    return out


def scale_cols(x2d, colscale):
    _require_cuda(x2d, colscale)
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    _hip.check(_hip.lib().iseg_scale_cols(ptr(x2d), ptr(colscale), ptr(y), rows, Cc, dt(x2d), stream()), "iseg_scale_cols")
    return y


def colsum_wide(x2d, out, accumulate=False):
    """out[c] (+)= sum_r x2d[r, c] for few rows and very many columns"""
    _require_cuda(x2d, out)
    rows, cols = x2d.shape
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_colsum_wide_workspace_bytes(rows, cols), x2d.device)
    _hip.check(L.iseg_colsum_wide(ptr(x2d), x2d.stride(0), rows, cols, ptr(out), int(accumulate), dt(x2d), ptr(ws), wsb, stream()),
               "iseg_colsum_wide")
    return out


# ---------------------------------------------------------------------------------------------------------
# fused window attention (csrc/winattn.hip)
# ---------------------------------------------------------------------------------------------------------
def attention_fwd_supported(head_dim, dtype):
    return dtype == torch.bfloat16 and bool(_hip.lib().iseg_attention_fwd_supported(int(head_dim), BF16))


def attention_fwd(qkv, heads, scale):
    """forward-only global attention on packed [q|k|v] (inference): [B, T, 3C] -> [B, T, C]"""
    B, T, ld = qkv.shape
    C = ld // 3
    out = torch.empty((B, T, C), dtype=qkv.dtype, device=qkv.device)
    _hip.check(_hip.lib().iseg_attention_fwd(ptr(qkv), ptr(out), B, T, heads, C // heads, float(scale), dt(qkv), stream()),
               "iseg_attention_fwd")
    return out


def attention_fwd_train(qkv, heads, scale):
    """attention_fwd that also returns the per-row log2-sum-exp vector the backward kernels recompute the probabilities from"""
    B, T, ld = qkv.shape
    C = ld // 3
    L = _hip.lib()
    out = torch.empty((B, T, C), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(int(L.iseg_attention_lse_elems(B, T, heads)), dtype=torch.float32, device=qkv.device)
    _hip.check(L.iseg_attention_fwd_train(ptr(qkv), ptr(out), ptr(lse), B, T, heads, C // heads, float(scale), dt(qkv), stream()),
               "iseg_attention_fwd_train")
    return out, lse


def attention_bwd(qkv, out, dout, lse, heads, scale):
    B, T, ld = qkv.shape
    C = ld // 3
    L = _hip.lib()
    dqkv = torch.empty_like(qkv)
    ws, wsb = workspace(L.iseg_attention_bwd_workspace_bytes(B, T, heads), qkv.device)
    _hip.check(L.iseg_attention_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), B, T, heads, C // heads, float(scale), dt(qkv),
                                    ptr(ws), wsb, stream()), "iseg_attention_bwd")
    return dqkv


def window_attention_supported(T, head_dim, dtype):
    return dtype == torch.bfloat16 and bool(_hip.lib().iseg_window_attention_supported(int(T), int(head_dim), BF16))


def window_attention_table(bias, mask, heads, T):
    """[mask windows (1 without mask), heads, 64, 64] fp32 additive table: bias + mask inside T x T, -FLT_MAX outside"""
    _require_cuda(bias)
    nW = mask.shape[0] if mask is not None else 1
    table = torch.empty((nW, heads, 64, 64), dtype=torch.float32, device=bias.device)
    _hip.check(_hip.lib().iseg_window_attention_table(ptr(bias), ptr(mask), ptr(table), T, heads, nW, stream()), "iseg_window_attention_table")
    return table


def window_attention_fwd(qkv, table, heads, scale):
    _require_cuda(qkv, table)
    B, T, ld = qkv.shape
    out = torch.empty((B, T, ld // 3), dtype=qkv.dtype, device=qkv.device)
    _hip.check(_hip.lib().iseg_window_attention_fwd(ptr(qkv), ptr(table), ptr(out), B, T, heads, table.shape[0], float(scale), dt(qkv), stream()),
               "iseg_window_attention_fwd")
    return out


def window_attention_bwd(qkv, table, dout, heads, scale):
    _require_cuda(qkv, table, dout)
    B, T, ld = qkv.shape
    dqkv = torch.empty_like(qkv)
    dbias = torch.empty((heads, T, T), dtype=torch.float32, device=qkv.device)
    L = _hip.lib()
    ws, wsb = workspace(L.iseg_window_attention_bwd_workspace_bytes(B, T, heads), qkv.device)
    _hip.check(L.iseg_window_attention_bwd(ptr(qkv), ptr(table), ptr(dout), ptr(dqkv), ptr(dbias), B, T, heads, table.shape[0], float(scale),
                                           dt(qkv), ptr(ws), wsb, stream()), "iseg_window_attention_bwd")
    return dqkv, dbias


# ---------------------------------------------------------------------------------------------------------
# deferred parameter-gradient reductions (csrc/api.hip): one batched launch instead of ~70 five-microsecond ones per step
# ---------------------------------------------------------------------------------------------------------
_DEFER = {"arena": None, "active": False}


@contextlib.contextmanager
def deferred_reductions(flat_grad, arena_bytes=96 << 20):
    """Between enter and exit, the second stage of LayerNorm / depthwise / bias gradient reductions whose output lies in `flat_grad` (and
    accumulates into it) is queued inside the library and run by ONE launch per flush.  Exit flushes; call deferred_flush() before
    anything else reads the gradient buffer (all-reduce, clipping).  ISEG_DEFER_REDUCE=0 turns the service off."""
    import os

    if not flat_grad.is_cuda or os.environ.get("ISEG_DEFER_REDUCE", "1") == "0" or _DEFER["active"]:
        yield
        return
    arena = _DEFER["arena"]
    if arena is None or arena.device != flat_grad.device or arena.numel() < arena_bytes:
        arena = _DEFER["arena"] = torch.empty(arena_bytes, dtype=torch.uint8, device=flat_grad.device)
    _hip.call("iseg_deferred_begin", ptr(flat_grad), flat_grad.numel() * flat_grad.element_size(), ptr(arena), arena.numel(), stream())
    _DEFER["active"] = True
    try:
        yield
    finally:
        _DEFER["active"] = False
        _hip.call("iseg_deferred_end", stream())


def deferred_flush():
    if _DEFER["active"]:
        _hip.call("iseg_deferred_flush", stream())
