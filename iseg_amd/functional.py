"""Differentiable operators: explicit forward/backward kernel pairs behind torch.autograd.Function.

torch.autograd only keeps the tape (which Function ran, what it saved); every forward and backward body is a
sequence of libiseg_hip.so calls.  Parameter gradients are written by the kernels straight into `param.grad`
(a view of the flat gradient buffer, see param_store.py) -- the Functions return None for parameters, so autograd
never runs an accumulation kernel of its own for them.
"""
import os

import torch
from torch.autograd import Function

from . import kernels as K
from . import nn
from . import dist


def _grad(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p.data)
    return p.grad


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _rows2d(t):
    """[..., C] as ([rows, C] view, row stride in elements) WITHOUT a copy when the rows are equally spaced and 16-byte aligned -- the
    gradient of a channel slice of a wider NHWC tensor (ASPP's concatenation) -- else through a contiguous copy"""
    C = t.shape[-1]
    if t.is_contiguous():
        return t.reshape(-1, C), C
    per16 = 16 // t.element_size()
    if t.dim() >= 2 and t.stride(-1) == 1 and all(t.stride(i) == t.stride(i + 1) * t.shape[i + 1] for i in range(t.dim() - 2)):
        ld = t.stride(-2)
        if ld % per16 == 0 and t.storage_offset() % per16 == 0 and C % per16 == 0:
            return t.as_strided((t.numel() // C, C), (ld, 1)), ld
    t = t.contiguous()
    return t.reshape(-1, C), C


def _dry(shape, like, dtype=None):
    return torch.empty(tuple(int(v) for v in shape), dtype=dtype or like.dtype, device=like.device)


def _check_act_dtype(x):
    if x.dtype != nn.compute_dtype():
        raise TypeError(f"activation dtype {x.dtype} != compute dtype {nn.compute_dtype()}; cast the input with F.cast_input")


# ---------------------------------------------------------------------------------------------------------
# input cast (image fp32 -> compute dtype); no gradient
# ---------------------------------------------------------------------------------------------------------
def cast_input(x):
    if x.dtype == nn.compute_dtype():
        return x
    if nn.dry_run():
        return _dry(x.shape, x, nn.compute_dtype())
    return K.cast(_c(x), nn.compute_dtype())


def cast_to(x, dtype):
    if nn.dry_run():
        return _dry(x.shape, x, dtype)
    return _CastFn.apply(x, dtype)


class _CastFn(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src_dtype = x.dtype
        return x if x.dtype == dtype else K.cast(_c(x), dtype)

    @staticmethod
    def backward(ctx, dy):
        return (dy if dy.dtype == ctx.src_dtype else K.cast(_c(dy), ctx.src_dtype)), None


# ---------------------------------------------------------------------------------------------------------
# Dense / 1x1 conv:  y = act(x @ W + b)         keras.layers.Dense, Conv2D(1x1)
# ---------------------------------------------------------------------------------------------------------
_FWD_KCONTIG = os.environ.get("ISEG_FWD_KCONTIG", "1") != "0"      # experiment knob: 0 = forward products read the Keras [K][N] kernels
_DMA_MIN_K = max(16, int(os.environ.get("ISEG_GEMM_DMA_MIN_K", "32")))      # the same knob csrc/gemm.hip reads (dma_min_k)


def _kcontig_kernel(W, x2, Kd, N):
    """the [N][K] copy of a forward product's kernel (nn.wt) when the LDS-DMA GEMM can take the product (csrc/gemm_dma.h dma_eligible: bf16,
    K a multiple of 8 and >= 32, N a multiple of 8 and >= 64, M >= 64), else None: the register-staged kernel reads [K][N] as it lies"""
    if not _FWD_KCONTIG or x2.dtype != torch.bfloat16 or Kd % 8 or Kd < _DMA_MIN_K or N % 8 or N < 64 or x2.shape[0] < 64:
        return None
    return nn.wt(W, (Kd, N))


class _DenseFn(Function):
    @staticmethod
    def forward(ctx, x, W, b, act, kshape=None):
        Kd, N = kshape if kshape is not None else (W.shape[-2], W.shape[-1])
        x2 = _c(x).reshape(-1, Kd)
        bias = b.data.reshape(-1) if b is not None else None
        pre = None
        need_grad = any(ctx.needs_input_grad)     # grad mode is off inside Function.forward; this is the tape's view
        if act == K.ACT_GELU and need_grad:
            pre = torch.empty((x2.shape[0], N), dtype=x2.dtype, device=x2.device)
        Wt = _kcontig_kernel(W, x2, Kd, N)
        if Wt is not None:
            y = K.dense_fwd_t(x2, Wt, bias, act=act, pre_out=pre)
        else:
            y = K.dense_fwd(x2, nn.w(W).reshape(Kd, N), bias, act=act, pre_out=pre)
        ctx.act, ctx.W, ctx.b, ctx.kshape = act, W, b, (Kd, N)
        ctx.save_for_backward(x2, pre if act == K.ACT_GELU else (y if act == K.ACT_RELU else None))
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, aux = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        Kd, N = ctx.kshape
        dy2 = _c(dy).reshape(-1, N)
        if ctx.act in (K.ACT_GELU, K.ACT_RELU):
            dy2 = K.act_bwd(dy2, aux, ctx.act)
        want_b = b is not None and b.requires_grad
        if W.requires_grad:
            # the bias gradient rides the weight-gradient GEMM (virtual ones-row) whenever the last 128-row tile has a spare row
            K.dense_wgrad(x2, dy2, _grad(W).reshape(Kd, N), bias_grad=(_grad(b).reshape(-1) if want_b else None))
        elif want_b:
            K.colsum(dy2, N, 0, 1, dy2.shape[0], N, _grad(b).reshape(-1), accumulate=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.dense_dgrad(dy2, nn.w(W).reshape(Kd, N)).reshape(*dy.shape[:-1], Kd)
        dist.grads_ready(W, b)
        return dx, None, None, None, None


class _MlpGeluFn(Function):
    """Dense -> exact GELU -> Dense of the transformer MLPs (backbones/swin.py:17-43 Mlp, backbones/vit.py:66-113 MLPBlock,
    backbones/intern_image/mlp_layer.py:48-59) as one tape node: the first GEMM's epilogue writes gelu(h) and gelu'(h), the second
    GEMM's data gradient multiplies by the saved derivative in its epilogue -- no elementwise activation-gradient pass."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, residual=None, rowscale=None):
        """residual / rowscale: `residual + rowscale[sample] * mlp(x)` -- the block's drop-path factor and skip connection ride the second
        product's epilogue (backbones/swin.py:233-236, backbones/vit.py:128-131) instead of a row-scale and an add kernel"""
        C, Hd = W1.shape[-2], W1.shape[-1]
        N = W2.shape[-1]
        x2 = _c(x).reshape(-1, C)
        res2 = _c(residual).reshape(-1, N) if residual is not None else None
        rpg = x2.shape[0] // rowscale.shape[0] if rowscale is not None else 0
        ctx.rowscale, ctx.rpg = rowscale, rpg
        need_grad = any(ctx.needs_input_grad)
        d = torch.empty((x2.shape[0], Hd), dtype=x2.dtype, device=x2.device) if need_grad else None
        W1t, W2t = _kcontig_kernel(W1, x2, C, Hd), None
        if W1t is not None:
            g = K.dense_fwd_t(x2, W1t, b1.data if b1 is not None else None, act=K.ACT_GELU, pre_out=d, pre_deriv=need_grad)
            W2t = _kcontig_kernel(W2, g, Hd, N)
        else:
            g = K.dense_fwd(x2, nn.w(W1), b1.data if b1 is not None else None, act=K.ACT_GELU, pre_out=d, pre_deriv=need_grad)
        if W2t is not None:
            y = K.dense_fwd_t(g, W2t, b2.data if b2 is not None else None, rowscale=rowscale, rows_per_group=rpg, residual=res2)
        else:
            y = K.dense_fwd(g, nn.w(W2), b2.data if b2 is not None else None, rowscale=rowscale, rows_per_group=rpg, residual=res2)
        ctx.params = (W1, b1, W2, b2)
        ctx.save_for_backward(x2, g if need_grad else None, d)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, g, d = ctx.saved_tensors
        W1, b1, W2, b2 = ctx.params
        N = W2.shape[-1]
        dres = dy if ctx.needs_input_grad[5] else None      # the skip connection passes the gradient through
        dy2 = _c(dy).reshape(-1, N)
        if ctx.rowscale is not None:
            dy2 = K.rowscale(dy2, ctx.rowscale, ctx.rpg)
        if W2.requires_grad:
            K.dense_wgrad(g, dy2, _grad(W2), bias_grad=(_grad(b2) if b2 is not None and b2.requires_grad else None))
        elif b2 is not None and b2.requires_grad:
            K.colsum(dy2, N, 0, 1, dy2.shape[0], N, _grad(b2), accumulate=True)
        dh = K.dense_dgrad(dy2, nn.w(W2), act=K.ACT_MUL_AUX, aux=d)            # (dy W2^T) * gelu'(h)
        if W1.requires_grad:
            K.dense_wgrad(x2, dh, _grad(W1), bias_grad=(_grad(b1) if b1 is not None and b1.requires_grad else None))
        elif b1 is not None and b1.requires_grad:
            K.colsum(dh, dh.shape[1], 0, 1, dh.shape[0], dh.shape[1], _grad(b1), accumulate=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.dense_dgrad(dh, nn.w(W1)).reshape(*dy.shape[:-1], W1.shape[-2])
        dist.grads_ready(W1, b1, W2, b2)
        return dx, None, None, None, None, dres, None


def mlp_gelu(x, W1, b1, W2, b2, residual=None, drop_path_mask=None):
    """dense(gelu(dense(x, W1, b1)), W2, b2) with no dropout in between; with `residual`: residual + drop_path_mask[sample] * that (the
    per-sample factors of utils/drops.py:8-22, None = no drop path), in the second product's epilogue"""
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry((*x.shape[:-1], W2.shape[-1]), x)
    if residual is None and drop_path_mask is not None:
        raise ValueError("mlp_gelu: drop_path_mask rides the residual epilogue")
    return _MlpGeluFn.apply(x, W1, b1, W2, b2, residual, drop_path_mask)


class _DenseGroupFn(Function):
    """Several Dense layers of ONE input whose outputs are wanted side by side -- keras MultiHeadAttention's query / key / value projections
    (backbones/vit.py:142-147: three kernels [C, heads, d]) feeding the packed attention kernels: every product writes its column block of
    the [M, sum N] result directly (no concat copy), and the backward pass reads the gradient's column blocks in place (strided operands: no
    slice copies) and accumulates the three data gradients in the GEMM epilogue (no adds)."""

    @staticmethod
    def forward(ctx, x, n, *wb):
        Ws, bs = wb[:n], wb[n:2 * n]
        Kd = x.shape[-1]
        x2 = _c(x).reshape(-1, Kd)
        Ns = [W.numel() // Kd for W in Ws]
        out = torch.empty((x2.shape[0], sum(Ns)), dtype=x2.dtype, device=x2.device)
        off = 0
        for W, b, N in zip(Ws, bs, Ns):
            view = out[:, off:off + N]
            bias = b.data.reshape(-1) if b is not None else None
            Wt = _kcontig_kernel(W, x2, Kd, N)
            if Wt is not None:
                K.dense_fwd_t(x2, Wt, bias, out=view)
            else:
                K.dense_fwd(x2, nn.w(W).reshape(Kd, N), bias, out=view)
            off += N
        ctx.Ws, ctx.bs, ctx.Ns, ctx.Kd = Ws, bs, Ns, Kd
        ctx.save_for_backward(x2)
        return out.reshape(*x.shape[:-1], sum(Ns))

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        Kd = ctx.Kd
        d2 = _c(dy).reshape(x2.shape[0], sum(ctx.Ns))
        dx, off = None, 0
        for W, b, N in zip(ctx.Ws, ctx.bs, ctx.Ns):
            dv = d2[:, off:off + N]      # a strided view: the GEMMs take the row stride
            want_b = b is not None and b.requires_grad
            if W.requires_grad:
                K.dense_wgrad(x2, dv, _grad(W).reshape(Kd, N), bias_grad=(_grad(b).reshape(-1) if want_b else None))
            elif want_b:
                K.colsum(dv, dv.stride(0), 0, 1, dv.shape[0], N, _grad(b).reshape(-1), accumulate=True)
            if ctx.needs_input_grad[0]:
                Wn = nn.w(W).reshape(Kd, N)
                dx = K.dense_dgrad(dv, Wn) if dx is None else K.dense_dgrad(dv, Wn, out=dx, residual=dx)      # + the earlier blocks, in the epilogue
            off += N
        dist.grads_ready(*ctx.Ws, *[b for b in ctx.bs if b is not None])
        return (dx.reshape(*dy.shape[:-1], Kd) if dx is not None else None, None) + (None,) * (2 * len(ctx.Ws))


def dense_group(x, kernels, biases):
    """[dense(x, W_i, b_i) for i] concatenated along the last axis as one tape node; a kernel of rank > 2 is read as [in, rest] (keras
    MultiHeadAttention: [C, heads, d]); biases: a list of the same length (entries may be None)"""
    _check_act_dtype(x)
    Kd = x.shape[-1]
    if nn.dry_run():
        return _dry((*x.shape[:-1], sum(W.numel() // Kd for W in kernels)), x)
    return _DenseGroupFn.apply(x, len(kernels), *kernels, *biases)


class _LnMlpResidualFn(Function):
    """x + drop_path(Dense(gelu(Dense(LayerNorm(x))))) -- the second half of a pre-norm transformer block (backbones/swin.py:233-236) -- on the
    fused kernels of the ConvNeXt stages (csrc/mlp_fused.hip, csrc/mlp_wgrad.hip: C = 96 / 192, hidden 4C, bf16): LayerNorm rides the row
    loads, the [M, 4C] hidden tile never leaves the CU in either direction, the skip connection and the drop-path factor ride the epilogue.
    Replaces LayerNorm + two GEMMs forward and LayerNorm backward + four GEMMs + their split-K sums + the residual fork backward."""

    @staticmethod
    def forward(ctx, x, ln_gamma, ln_beta, eps, W1, b1, W2, b2, rowscale):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        M = x2.shape[0]
        rpg = M // rowscale.shape[0] if rowscale is not None else 0
        fw, bw = nn.mlp_tiled(W1, W2, None)
        out, mean, rstd = K.convnext_mlp_fwd_ln(x2, ln_gamma.data, ln_beta.data, eps, fw, b1.data, b2.data, None, rowscale, rpg, x2)
        ctx.params, ctx.rpg = (ln_gamma, ln_beta, W1, b1, W2, b2), rpg
        ctx.save_for_backward(x2, mean, rstd, bw, rowscale)
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x2, mean, rstd, bw, rowscale = ctx.saved_tensors
        ln_gamma, ln_beta, W1, b1, W2, b2 = ctx.params
        M, C = x2.shape
        do2 = _c(dout).reshape(M, C)
        ln = (mean, rstd, ln_gamma.data, ln_beta.data)
        dx = None
        if ctx.needs_input_grad[0]:
            dln = K.convnext_mlp_bwd_data_ln(x2, do2, bw, b1.data, ln, _grad(ln_gamma), _grad(ln_beta), rowscale, ctx.rpg)
            dx = K.axpby(do2, dln, 1.0, 1.0, out=dln).reshape(dout.shape)      # + the skip connection's share
        else:      # (the LayerNorm parameter gradients come out of the chain kernel's epilogue)
            K.convnext_mlp_bwd_data_ln(x2, do2, bw, b1.data, ln, _grad(ln_gamma), _grad(ln_beta), rowscale, ctx.rpg)
        K.convnext_mlp_wgrad(x2, do2, bw, b1.data, W2.data, b2.data, None, _grad(W1), _grad(b1), _grad(W2), _grad(b2), None, rowscale, ctx.rpg, ln=ln)
        dist.grads_ready(ln_gamma, ln_beta, W1, b1, W2, b2)
        return (dx,) + (None,) * 8


def ln_mlp_residual_supported(x, params, drop_path_mask=None):
    """the fused route exists for bf16 rows of 96 / 192 channels with a 4x hidden layer, every parameter trainable (or none needed), and a
    drop-path group size the kernels' 64-row tiles divide"""
    ln_gamma, ln_beta, W1, b1, W2, b2 = params
    C = x.shape[-1]
    if any(p is None for p in params) or not K.convnext_mlp_supported(C, x.dtype) or C not in (96, 192):
        return False
    if tuple(W1.shape) != (C, 4 * C) or tuple(W2.shape) != (4 * C, C):
        return False
    if not all(p.requires_grad for p in params) and torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        return False      # partly frozen: the weight-gradient kernel books all six parameters
    rows = x.numel() // C
    return drop_path_mask is None or (rows % drop_path_mask.shape[0] == 0 and (rows // drop_path_mask.shape[0]) % 64 == 0)


def ln_mlp_residual(x, ln_gamma, ln_beta, eps, W1, b1, W2, b2, drop_path_mask=None):
    """x + drop_path_mask[sample] * dense(gelu(dense(layer_norm(x)))) as ONE tape node (ln_mlp_residual_supported(...) must hold)"""
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    return _LnMlpResidualFn.apply(x, ln_gamma, ln_beta, float(eps), W1, b1, W2, b2, drop_path_mask)


def dense(x, W, b=None, act=K.ACT_NONE, kshape=None):
    """kshape=(in, out) re-interprets a higher-rank kernel (keras MultiHeadAttention: [C, heads, d] / [heads, d, C]) as [in, out]"""
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry((*x.shape[:-1], kshape[1] if kshape is not None else W.shape[-1]), x)
    return _DenseFn.apply(x, W, b, act, kshape)


# ---------------------------------------------------------------------------------------------------------
# Conv2D;   kernel [kh,kw,Cin/groups,Cout]
# ---------------------------------------------------------------------------------------------------------
def _conv_geometry(H, W, kh, kw, strides, dilation, padding):
    sh, sw = strides
    dh, dw = dilation
    if padding == "same":
        Ho, pt = K.same_pad(H, kh, sh, dh)
        Wo, pl = K.same_pad(W, kw, sw, dw)
    elif padding == "valid":
        Ho = (H - (kh - 1) * dh - 1) // sh + 1
        Wo = (W - (kw - 1) * dw - 1) // sw + 1
        pt = pl = 0
    elif isinstance(padding, tuple):          # ((top, bottom), (left, right)): ZeroPadding2D followed by a "valid" window op
        (pt, pb), (pl, pr) = padding
        Ho = (H + pt + pb - (kh - 1) * dh - 1) // sh + 1
        Wo = (W + pl + pr - (kw - 1) * dw - 1) // sw + 1
    else:
        raise ValueError(f"padding {padding!r} not supported")
    return Ho, Wo, pt, pl


_IGEMM_PHASES = os.environ.get("ISEG_IGEMM_PHASES", "1") != "0"      # experiment knob: 0 = strided data gradients through GEMM + col2im


class _Conv2dFn(Function):
    """Three routes: (1) 1x1 / stride 1 / one group on the activation as it lies = a plain GEMM; (2) bf16 storage with channels per
    group that are multiples of 8 = implicit GEMM on the matrix cores (csrc/conv_igemm.hip: the patch matrix is gathered while the
    operand tile is staged, the data gradient is a gather over dy -- no column buffer, no col2im); (3) everything else (the fp32 parity
    mode, Cin = 3 stems) = im2col + GEMM + col2im, group by group."""

    @staticmethod
    def forward(ctx, x, W, b, strides, dilation, padding, groups):
        kh, kw, Cin_g, Cout = W.shape
        N, H, Wd, Cin = x.shape
        if Cin != Cin_g * groups or Cout % groups != 0:
            raise ValueError(f"Conv2D: kernel {tuple(W.shape)} does not fit {Cin} input channels in {groups} groups")
        Ho, Wo, pt, pl = _conv_geometry(H, Wd, kh, kw, strides, dilation, padding)
        cdt = nn.compute_dtype()
        xc = _c(x)
        geom = K.conv_geom(N, H, Wd, Cin, Cout, kh, kw, strides[0], strides[1], dilation[0], dilation[1], pt, pl, Ho, Wo, groups)
        direct = kh == 1 and kw == 1 and strides == (1, 1) and groups == 1 and xc.dtype == cdt and Cin % 8 == 0
        igemm = (not direct) and xc.dtype == cdt and K.conv2d_igemm_supported(geom, cdt)
        M = N * Ho * Wo
        if direct:
            Wt = _kcontig_kernel(W, xc.reshape(-1, Cin), Cin, Cout)
            if Wt is not None:
                y = K.dense_fwd_t(xc.reshape(-1, Cin), Wt, b.data if b is not None else None)
            else:
                y = torch.empty((M, Cout), dtype=cdt, device=x.device)
                K.gemm(xc.reshape(-1, Cin), nn.w(W).reshape(Cin, Cout), y, M, Cout, Cin, lda=Cin, ldb=Cout, ldd=Cout, a_kcontig=1, b_kcontig=0,
                       bias=(b.data if b is not None else None))
        elif igemm:
            Wt = None
            if _FWD_KCONTIG and K.conv2d_igemm_fwd_kt_supported(geom, cdt):      # LDS-DMA form on the K-contiguous kernel copy
                Wt = nn.wt(W, (kh * kw * Cin, Cout))
            if Wt is not None:
                y = K.conv2d_igemm_fwd_kt(xc, Wt, b.data if b is not None else None, geom)
            else:
                y = K.conv2d_igemm_fwd(xc, nn.w(W), b.data if b is not None else None, geom)
        else:
            y = torch.empty((M, Cout), dtype=cdt, device=x.device)
            Kd, og = kh * kw * Cin_g, Cout // groups
            keep_col = None
            for g in range(groups):
                col = K.im2col(_group_slice(xc, g, Cin_g, groups), kh, kw, strides[0], strides[1], dilation[0], dilation[1], pt, pl, Ho, Wo, cdt)
                K.gemm(col, nn.w(W).reshape(Kd, Cout)[:, g * og:], y[:, g * og:], M, og, Kd, lda=col.stride(0), ldb=Cout, ldd=Cout, a_kcontig=1,
                       b_kcontig=0, bias=(b.data[g * og:(g + 1) * og] if b is not None else None))
                keep_col = col
            # a patchify stem on the image (Cin = 3, kernel == stride): the column buffer is smaller than the fp32 image it came from, and
            # the weight gradient is its only other reader -- keep it instead of running im2col again in backward (63 us at 16 x 512 x 512)
            ctx.col = keep_col if (groups == 1 and W.requires_grad and not ctx.needs_input_grad[0] and Kd * keep_col.element_size() <=
                                   Cin * xc.element_size() * strides[0] * strides[1]) else None
        ctx.W, ctx.b = W, b
        ctx.geom, ctx.route = geom, (direct, igemm)
        ctx.x_dtype = x.dtype
        ctx.save_for_backward(xc)
        return y.reshape(N, Ho, Wo, Cout)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        W, b, g_ = ctx.W, ctx.b, ctx.geom
        direct, igemm = ctx.route
        N, H, Wd, Cin, Cout, kh, kw, groups = g_.N, g_.H, g_.W, g_.Cin, g_.Cout, g_.KH, g_.KW, g_.groups
        Ho, Wo, pt, pl = g_.Ho, g_.Wo, g_.pt, g_.pl
        st, di = (g_.sh, g_.sw), (g_.dh, g_.dw)
        Cin_g, og = Cin // groups, Cout // groups
        Kd = kh * kw * Cin_g
        cdt = nn.compute_dtype()
        M = N * Ho * Wo
        dy2 = _c(dy).reshape(M, Cout)
        if b is not None and b.requires_grad:
            K.colsum(dy2, Cout, 0, 1, M, Cout, _grad(b), accumulate=True)
        need_dx = ctx.needs_input_grad[0]
        dx = None
        if direct:
            if W.requires_grad:
                K.gemm(xc.reshape(-1, Cin), dy2, _grad(W).reshape(Cin, Cout), Cin, Cout, M, lda=Cin, ldb=Cout, ldd=Cout, a_kcontig=0, b_kcontig=0,
                       accumulate=True)
            if need_dx:
                dx = K.dense_dgrad(dy2, nn.w(W).reshape(Cin, Cout)).reshape(N, H, Wd, Cin)
        elif igemm:
            dy4 = dy2.reshape(N, Ho, Wo, Cout)
            if W.requires_grad:
                K.conv2d_igemm_bwd_weight(xc, dy4, _grad(W), g_, accumulate=True)
            patchify = (kh, kw) == st      # kernel == stride: the column buffer IS dx up to a permutation, one plain GEMM fills it
            if need_dx and (st == (1, 1) or (di == (1, 1) and st[0] * st[1] <= 16 and _IGEMM_PHASES and not patchify)):
                # stride 1: one gather over dy; strided and undilated: one stride-1 gather per stride phase, all phases in one launch, rows
                # scattered to their pixels by the epilogue (csrc/conv_igemm.hip pass 3) -- no zero products, no column buffer, no col2im
                # (ResNet's 3x3 / s2 at 64x64x128: 23 us against 40 us for GEMM + col2im)
                dx = K.conv2d_igemm_bwd_data(dy4, nn.w(W), g_)
            elif need_dx:
                # strided and dilated (1 / (sh * sw) of the (tap, pixel) pairs of the plain gather form are non-zero, and the MFMA cannot skip
                # them), or kernel == stride (ConvNeXt's 2x2 / s2 downsamples: the LDS-DMA GEMM + permutation measures 35 / 29 us against
                # 46 / 39 us for the phase launch at stages 1 / 2, equal at stage 0; tools/kbench_conv_strided.py): dcol = dy @ W^T, then the
                # gather-form col2im (deterministic)
                ldc = (Kd + 7) // 8 * 8
                dx = torch.empty((N, H, Wd, Cin), dtype=cdt, device=dy.device) if groups > 1 else None
                for g in range(groups):
                    dcol = torch.empty((M, ldc), dtype=cdt, device=dy.device)
                    K.gemm(dy2[:, g * og:], nn.w(W).reshape(Kd, Cout)[:, g * og:], dcol, M, Kd, og, lda=Cout, ldb=Cout, ldd=ldc, a_kcontig=1,
                           b_kcontig=1)
                    dxg = K.col2im(dcol, N, H, Wd, Cin_g, kh, kw, st[0], st[1], di[0], di[1], pt, pl, Ho, Wo)
                    if groups == 1:
                        dx = dxg
                    else:
                        K.copy2d(dxg.reshape(-1, Cin_g), Cin_g, dx.reshape(-1, Cin)[:, g * Cin_g:], Cin, N * H * Wd, Cin_g)
        else:
            if need_dx:
                dx = torch.empty((N, H, Wd, Cin), dtype=cdt, device=dy.device)
            ldc = (Kd + 7) // 8 * 8
            for g in range(groups):
                dyg = dy2[:, g * og:]
                if W.requires_grad:
                    col = getattr(ctx, "col", None)
                    if col is None:
                        col = K.im2col(_group_slice(xc, g, Cin_g, groups), kh, kw, st[0], st[1], di[0], di[1], pt, pl, Ho, Wo, cdt)
                    K.gemm(col, dyg, _grad(W).reshape(Kd, Cout)[:, g * og:], Kd, og, M, lda=col.stride(0), ldb=Cout, ldd=Cout, a_kcontig=0,
                           b_kcontig=0, accumulate=True)
                    del col
                if need_dx:
                    dcol = torch.empty((M, ldc), dtype=cdt, device=dy.device)
                    K.gemm(dyg, nn.w(W).reshape(Kd, Cout)[:, g * og:], dcol, M, Kd, og, lda=Cout, ldb=Cout, ldd=ldc, a_kcontig=1, b_kcontig=1)
                    dxg = K.col2im(dcol, N, H, Wd, Cin_g, kh, kw, st[0], st[1], di[0], di[1], pt, pl, Ho, Wo)
                    if groups == 1:
                        dx = dxg
                    else:
                        K.copy2d(dxg.reshape(-1, Cin_g), Cin_g, dx.reshape(-1, Cin)[:, g * Cin_g:], Cin, N * H * Wd, Cin_g)
        if dx is not None and dx.dtype != ctx.x_dtype:
            dx = K.cast(dx, ctx.x_dtype)
        dist.grads_ready(W, b)
        return dx, None, None, None, None, None, None


def _group_slice(xc, g, Cin_g, groups):
    """dense NHWC copy of one channel group (the im2col kernel reads dense tensors); groups == 1 is the tensor itself"""
    if groups == 1:
        return xc
    N, H, W, Cin = xc.shape
    out = torch.empty((N, H, W, Cin_g), dtype=xc.dtype, device=xc.device)
    K.copy2d(xc.reshape(-1, Cin)[:, g * Cin_g:], Cin, out.reshape(-1, Cin_g), Cin_g, N * H * W, Cin_g)
    return out


def conv2d(x, W, b=None, strides=(1, 1), dilation=(1, 1), padding="same", groups=1):
    if nn.dry_run():
        Ho, Wo, _, _ = _conv_geometry(x.shape[1], x.shape[2], W.shape[0], W.shape[1], tuple(strides), tuple(dilation), padding)
        return _dry((x.shape[0], Ho, Wo, W.shape[-1]), x, nn.compute_dtype())
    return _Conv2dFn.apply(x, W, b, tuple(strides), tuple(dilation), padding, int(groups))


# ---------------------------------------------------------------------------------------------------------
# DepthwiseConv2D, stride 1;  kernel [K,K,C,1]
# ---------------------------------------------------------------------------------------------------------
class _DWConvFn(Function):
    @staticmethod
    def forward(ctx, x, W, b, dil):
        Kk, C = W.shape[0], W.shape[2]
        pad = (Kk - 1) * dil // 2
        xc = _c(x)
        y = K.dwconv2d(xc, W.data.reshape(Kk * Kk, C), b.data if b is not None else None, Kk, dil, pad, pad)
        ctx.W, ctx.b, ctx.dil, ctx.pad = W, b, dil, pad
        ctx.save_for_backward(xc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        W, b, dil, pad = ctx.W, ctx.b, ctx.dil, ctx.pad
        Kk, C = W.shape[0], W.shape[2]
        dyc = _c(dy)
        if W.requires_grad:
            K.dwconv2d_bwd_weight(xc, dyc, _grad(W).reshape(Kk * Kk, C), _grad(b) if b is not None else None, Kk, dil, pad, pad)
        dx = None
        if ctx.needs_input_grad[0]:
            padb = (Kk - 1) * dil - pad
            dx = K.dwconv2d(dyc, W.data.reshape(Kk * Kk, C), None, Kk, dil, padb, padb, flip=True)
        dist.grads_ready(W, b)
        return dx, None, None, None


class _DWConvStridedFn(Function):
    """DepthwiseConv2D(strides = s, padding = 'same'): forward, data gradient (gather over dy) and weight gradient at the strided output
    positions only (csrc/dwconv_strided.hip)"""

    @staticmethod
    def forward(ctx, x, W, b, dil, stride):
        Kk, C = W.shape[0], W.shape[2]
        xc = _c(x)
        y = K.dwconv2d_strided(xc, W.data.reshape(Kk * Kk, C), b.data if b is not None else None, Kk, stride, dil)
        ctx.W, ctx.b, ctx.dil, ctx.stride = W, b, dil, stride
        ctx.save_for_backward(xc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        W, b, dil, stride = ctx.W, ctx.b, ctx.dil, ctx.stride
        Kk, C = W.shape[0], W.shape[2]
        dyc = _c(dy)
        if W.requires_grad:
            K.dwconv2d_strided_bwd_weight(xc, dyc, _grad(W).reshape(Kk * Kk, C), _grad(b) if b is not None else None, Kk, stride, dil)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.dwconv2d_strided_bwd_data(dyc, W.data.reshape(Kk * Kk, C), Kk, stride, dil, xc.shape[1], xc.shape[2])
        dist.grads_ready(W, b)
        return dx, None, None, None, None


def depthwise_conv2d(x, W, b=None, dilation=1, strides=1):
    """padding='same'.  Stride 1 is the hot path (ConvNeXt 7x7, SepConvBnReLU 3x3 dilated); a strided layer (the stride-2 depthwise
    convolutions of the separable / inverted-residual families) runs the dedicated kernels that visit the strided outputs only."""
    _check_act_dtype(x)
    s = int(strides)
    if s == 1:
        if nn.dry_run():
            return _dry(x.shape, x)
        return _DWConvFn.apply(x, W, b, int(dilation))
    N, H, Wd, C = x.shape
    if nn.dry_run():
        return _dry((N, -(-H // s), -(-Wd // s), C), x)
    return _DWConvStridedFn.apply(x, W, b, int(dilation), s)


# ---------------------------------------------------------------------------------------------------------
# LayerNormalization(axis=-1)
# ---------------------------------------------------------------------------------------------------------
class _LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        y, mean, rstd = K.layernorm_fwd(x2, gamma.data, beta.data, eps)
        ctx.gamma, ctx.beta = gamma, beta
        ctx.save_for_backward(x2, mean, rstd)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        C = x2.shape[-1]
        dx = K.layernorm_bwd(_c(dy).reshape(-1, C), x2, ctx.gamma.data, mean, rstd, _grad(ctx.gamma), _grad(ctx.beta))
        dist.grads_ready(ctx.gamma, ctx.beta)
        return dx.reshape(dy.shape), None, None, None


class _LayerNormPostFn(Function):
    """residual + drop_path_mask[sample] * colscale * layer_norm(x): the tail of a post-norm residual branch (InternImage, post_norm = True) as one
    pass each way -- LayerNorm, layer scale, drop path and the skip connection forward; backward one LayerNorm-backward pass whose two column
    sums also give the layer-scale gradient (csrc/norm.hip LnPost)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, colscale, rowscale, residual):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        rpg = x2.shape[0] // rowscale.shape[0] if rowscale is not None else 0
        r2 = _c(residual).reshape(-1, C) if residual is not None else None
        y, mean, rstd = K.layernorm_post_fwd(x2, gamma.data, beta.data, eps, colscale.data if colscale is not None else None, rowscale, rpg, r2)
        ctx.params, ctx.rpg = (gamma, beta, colscale), rpg
        ctx.save_for_backward(x2, mean, rstd, rowscale)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, rowscale = ctx.saved_tensors
        gamma, beta, colscale = ctx.params
        C = x2.shape[1]
        want_cs = colscale is not None and colscale.requires_grad
        dx = K.layernorm_post_bwd(_c(dy).reshape(-1, C), x2, gamma.data, beta.data, mean, rstd, _grad(gamma), _grad(beta),
                                  colscale=colscale.data if colscale is not None else None, dcolscale=_grad(colscale) if want_cs else None,
                                  rowscale=rowscale, rows_per_group=ctx.rpg)
        dist.grads_ready(gamma, beta, colscale) if colscale is not None else dist.grads_ready(gamma, beta)
        return (dx.reshape(dy.shape) if ctx.needs_input_grad[0] else None), None, None, None, None, None, (dy if ctx.needs_input_grad[6] else None)


def layer_norm_post(x, gamma, beta, eps, colscale=None, drop_path_mask=None, residual=None):
    """residual + drop_path_mask[sample] * colscale * layer_norm(x, gamma, beta, eps) as one tape node (channels % 8 == 0; gamma and beta trainable)"""
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    return _LayerNormPostFn.apply(x, gamma, beta, float(eps), colscale, drop_path_mask, residual)


def layer_norm(x, gamma, beta, eps):
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    return _LayerNormFn.apply(x, gamma, beta, float(eps))


# ---------------------------------------------------------------------------------------------------------
# (Sync)BatchNormalization over all axes but the last, optional fused ReLU
# ---------------------------------------------------------------------------------------------------------
class _BatchNormTrainFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, relu, sync):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        rows = x2.shape[0]
        packed = K.bn_stats(x2, C, rows, C)
        if sync:
            dist.all_reduce_sum(packed)          # ONE [2C+1] message (sum, sumsq, count) instead of the reference's three
        y = torch.empty_like(x2)
        mean, rstd = K.bn_finalize_apply(packed, x2, C, gamma.data, beta.data, y, C, rows, C, eps, momentum, moving_mean, moving_var, relu)
        ctx.gamma, ctx.beta, ctx.relu, ctx.sync = gamma, beta, relu, sync
        ctx.save_for_backward(x2, y if relu else None, mean, rstd, packed)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, y, mean, rstd, packed = ctx.saved_tensors
        rows, C = x2.shape
        dy2, lddy = _rows2d(dy)
        sums = K.bn_bwd_reduce(dy2, lddy, x2, C, y, C, mean, rstd, rows, C, ctx.relu)
        dbeta = _grad(ctx.beta) if ctx.beta.requires_grad else None
        dgamma = _grad(ctx.gamma) if ctx.gamma.requires_grad else None
        dx = torch.empty_like(x2)
        if ctx.sync and dist.active():
            # local parameter gradients first (the gradient all-reduce sums them over ranks later), then the statistics of all replicas
            K.accumulate_pair(sums, C, dbeta, dgamma)
            dist.grads_ready(ctx.gamma, ctx.beta)
            sums = sums.clone()
            dist.all_reduce_sum(sums)
            K.bn_bwd_apply(dy2, lddy, x2, C, y, C, mean, rstd, ctx.gamma.data, sums, 1.0 / (rows * dist.world_size()), dx, C, rows, C, ctx.relu)
        else:      # the sums are this replica's own: the apply kernel books dbeta | dgamma itself
            K.bn_bwd_apply(dy2, lddy, x2, C, y, C, mean, rstd, ctx.gamma.data, sums, 1.0 / rows, dx, C, rows, C, ctx.relu, dgamma=dgamma,
                           dbeta=dbeta)
            dist.grads_ready(ctx.gamma, ctx.beta)
        return dx.reshape(dy.shape), None, None, None, None, None, None, None, None


class _BnReluUpsampleAddFn(Function):
    """relu(batch_norm(z)) + resize_bilinear(x, z's height and width) -- one level of the FPN top-down pathway (layers/fpn.py:46-57) -- with
    training-mode (Sync)BN statistics: the normalised map and the up-sampled map are never written, and the backward pass re-derives the
    ReLU mask from z instead of reading a saved output.  Statistics / parameter-gradient arithmetic is _BatchNormTrainFn's."""

    @staticmethod
    def forward(ctx, z, x, gamma, beta, moving_mean, moving_var, eps, momentum, sync):
        N, Ho, Wo, C = z.shape
        zc, xc = _c(z), _c(x)
        z2 = zc.reshape(-1, C)
        rows = z2.shape[0]
        packed = K.bn_stats(z2, C, rows, C)
        if sync:
            dist.all_reduce_sum(packed)
        mean, rstd = K.bn_finalize(packed, C, eps, momentum, moving_mean, moving_var)
        out = K.bn_relu_upsample_add(zc, mean, rstd, gamma.data, beta.data, xc)
        ctx.gamma, ctx.beta, ctx.sync, ctx.x_shape, ctx.x_dtype = gamma, beta, sync, x.shape, x.dtype
        ctx.save_for_backward(z2, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        z2, mean, rstd = ctx.saved_tensors
        rows, C = z2.shape
        dc = _c(dout)
        d2 = dc.reshape(rows, C)
        dx = K.resize_bilinear_bwd(dc, ctx.x_shape[1], ctx.x_shape[2], ctx.x_dtype) if ctx.needs_input_grad[1] else None
        dz = None
        g, b = ctx.gamma.data, ctx.beta.data
        sums = K.bn_bwd_reduce_remask(d2, C, z2, C, mean, rstd, g, b, rows, C)
        dbeta = _grad(ctx.beta) if ctx.beta.requires_grad else None
        dgamma = _grad(ctx.gamma) if ctx.gamma.requires_grad else None
        if ctx.sync and dist.active():
            K.accumulate_pair(sums, C, dbeta, dgamma)
            dist.grads_ready(ctx.gamma, ctx.beta)
            sums = sums.clone()
            dist.all_reduce_sum(sums)
            if ctx.needs_input_grad[0]:
                dz = K.bn_bwd_apply_remask(d2, C, z2, C, mean, rstd, g, b, sums, 1.0 / (rows * dist.world_size()), torch.empty_like(z2), C, rows, C)
        else:
            if ctx.needs_input_grad[0]:
                dz = K.bn_bwd_apply_remask(d2, C, z2, C, mean, rstd, g, b, sums, 1.0 / rows, torch.empty_like(z2), C, rows, C, dgamma=dgamma, dbeta=dbeta)
            else:
                K.accumulate_pair(sums, C, dbeta, dgamma)
            dist.grads_ready(ctx.gamma, ctx.beta)
        return (dz.reshape(dout.shape) if dz is not None else None), dx, None, None, None, None, None, None, None


def batch_norm_relu_upsample_add(z, x, gamma, beta, moving_mean, moving_var, eps, momentum, sync=True):
    """relu(batch_norm(z, training=True)) + resize_bilinear(x, z.shape[1:3]) as one tape node; channels % 8 == 0, same dtype"""
    _check_act_dtype(z)
    if nn.dry_run():
        return _dry(z.shape, z)
    return _BnReluUpsampleAddFn.apply(z, x, gamma, beta, moving_mean, moving_var, float(eps), float(momentum), bool(sync))


class _BatchNormGroupFn(Function):
    """Several independent SyncBN layers whose inputs all exist before any of them is normalised (the five ASPP branches,
    layers/aspp.py:57-71): their packed statistics travel in ONE all-reduce forward and ONE backward instead of one per layer --
    the exposed latency of a small RCCL message is paid once (hard part 4).  Arithmetic per layer is _BatchNormTrainFn's."""

    @staticmethod
    def forward(ctx, n, relu, layers, *tensors):
        xs, gammas, betas = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        x2s = [_c(x).reshape(-1, x.shape[-1]) for x in xs]
        Cs = [x2.shape[1] for x2 in x2s]
        offs = [0]
        for C in Cs:
            offs.append(offs[-1] + 2 * C + 4)      # [sum | sumsq | count] + 3 floats of padding: every layer's slot stays 16-byte aligned
        msg = torch.zeros(offs[-1], dtype=torch.float32, device=x2s[0].device)
        for i, x2 in enumerate(x2s):
            K.bn_stats(x2, Cs[i], x2.shape[0], Cs[i], out=msg[offs[i]:offs[i] + 2 * Cs[i] + 1])
        dist.all_reduce_sum(msg)
        ys, saved = [], []
        for i, x2 in enumerate(x2s):
            L = layers[i]
            y = torch.empty_like(x2)
            mean, rstd = K.bn_finalize_apply(msg[offs[i]:offs[i] + 2 * Cs[i] + 1], x2, Cs[i], gammas[i].data, betas[i].data, y, Cs[i], x2.shape[0], Cs[i],
                                             L["eps"], L["momentum"], L["moving_mean"], L["moving_var"], relu)
            ys.append(y.reshape(xs[i].shape))
            saved += [x2, y if relu else None, mean, rstd]
        ctx.n, ctx.relu, ctx.gammas, ctx.betas = n, relu, gammas, betas
        ctx.save_for_backward(*saved)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n, relu = ctx.n, ctx.relu
        sv = ctx.saved_tensors
        Cs = [sv[4 * i].shape[1] for i in range(n)]
        offs = [0]
        for C in Cs:
            offs.append(offs[-1] + 2 * C)
        msg = torch.empty(offs[-1], dtype=torch.float32, device=sv[0].device)
        dy2s = []
        for i in range(n):
            x2, y, mean, rstd = sv[4 * i:4 * i + 4]
            rows, C = x2.shape
            dy2, lddy = _rows2d(dys[i])
            dy2s.append((dy2, lddy))
            sums = K.bn_bwd_reduce(dy2, lddy, x2, C, y, C, mean, rstd, rows, C, relu, out=msg[offs[i]:offs[i + 1]])
            # local parameter gradients (the gradient all-reduce sums them over ranks later)
            K.accumulate_pair(sums, C, _grad(ctx.betas[i]) if ctx.betas[i].requires_grad else None,
                              _grad(ctx.gammas[i]) if ctx.gammas[i].requires_grad else None)
            dist.grads_ready(ctx.gammas[i], ctx.betas[i])
        world = 1
        if dist.active():
            msg = msg.clone()
            dist.all_reduce_sum(msg)
            world = dist.world_size()
        dxs = []
        for i in range(n):
            x2, y, mean, rstd = sv[4 * i:4 * i + 4]
            rows, C = x2.shape
            dx = torch.empty_like(x2)
            K.bn_bwd_apply(dy2s[i][0], dy2s[i][1], x2, C, y, C, mean, rstd, ctx.gammas[i].data, msg[offs[i]:offs[i + 1]], 1.0 / (rows * world),
                           dx, C, rows, C, relu)
            dxs.append(dx.reshape(dys[i].shape))
        return (None, None, None) + tuple(dxs) + (None,) * (2 * n)


def batch_norm_group(xs, bns, relu=True):
    """training-mode SyncBN of several tensors with one statistics exchange; bns: the BatchNormalization layers (built)"""
    layers = [dict(eps=float(b.epsilon), momentum=float(b.momentum), moving_mean=b.moving_mean, moving_var=b.moving_variance) for b in bns]
    return _BatchNormGroupFn.apply(len(xs), bool(relu), layers, *xs, *[b.gamma for b in bns], *[b.beta for b in bns])


class _BatchNormInferFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, relu):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        rstd = K.rsqrt_eps(moving_var, eps)
        y = torch.empty_like(x2)
        K.bn_apply_fwd(x2, C, moving_mean, rstd, gamma.data, beta.data, y, C, x2.shape[0], C, relu)
        ctx.gamma, ctx.beta, ctx.relu = gamma, beta, relu
        want_param_grads = gamma.requires_grad or beta.requires_grad
        ctx.save_for_backward(y if relu else None, rstd, x2 if want_param_grads else None, moving_mean if want_param_grads else None)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        # frozen statistics: dx = dz * gamma * rstd; gamma / beta still receive their gradients (Keras keeps them trainable
        # when a BN layer is called with training=False inside a train step)
        y, rstd, x2, mean = ctx.saved_tensors
        C = dy.shape[-1]
        dy2 = _c(dy).reshape(-1, C)
        if x2 is not None:
            sums = K.bn_bwd_reduce(dy2, C, x2, C, y, C, mean, rstd, x2.shape[0], C, ctx.relu)
            K.accumulate_pair(sums, C, _grad(ctx.beta) if ctx.beta.requires_grad else None,
                              _grad(ctx.gamma) if ctx.gamma.requires_grad else None)
            dist.grads_ready(ctx.gamma, ctx.beta)
        if ctx.relu:
            dy2 = K.act_bwd(dy2, y, K.ACT_RELU)
        scale = K.axpby(ctx.gamma.data, None, 1.0, 0.0)
        # dx = dy2 * (gamma*rstd) per column: reuse bn_apply with mean=0,rstd=1,beta=0, gamma=gamma*rstd
        zeros = torch.zeros(C, dtype=torch.float32, device=dy.device)
        ones = torch.ones(C, dtype=torch.float32, device=dy.device)
        dx = torch.empty_like(dy2)
        K.bn_apply_fwd(dy2, C, zeros, rstd, scale, zeros, dx, C, dy2.shape[0], C, False)
        del ones
        return dx.reshape(dy.shape), None, None, None, None, None, None


def batch_norm(x, gamma, beta, moving_mean, moving_var, eps, momentum, training, relu=False, sync=True):
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    if training:
        return _BatchNormTrainFn.apply(x, gamma, beta, moving_mean, moving_var, float(eps), float(momentum), bool(relu), bool(sync))
    return _BatchNormInferFn.apply(x, gamma, beta, moving_mean, moving_var, float(eps), bool(relu))


# ---------------------------------------------------------------------------------------------------------
# activations, add, dropout, drop-path
# ---------------------------------------------------------------------------------------------------------
class _ActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        xc = _c(x)
        y = K.act_fwd(xc, act)
        ctx.act = act
        ctx.save_for_backward(y if act == K.ACT_RELU else xc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (aux,) = ctx.saved_tensors
        return K.act_bwd(_c(dy), aux, ctx.act), None


def relu(x):
    if nn.dry_run():
        return _dry(x.shape, x)
    return _ActFn.apply(x, K.ACT_RELU)


def gelu(x):
    if nn.dry_run():
        return _dry(x.shape, x)
    return _ActFn.apply(x, K.ACT_GELU)


def sigmoid(x):
    """tf.nn.sigmoid (layers/nasfpn.py:304-311) through the C-ABI activation kernel"""
    if nn.dry_run():
        return _dry(x.shape, x)
    return _ActFn.apply(x, K.ACT_SIGMOID)


def swish(x):
    """tf.nn.silu / keras.activations.swish: x * sigmoid(x) (backbones/eva/swiglu.py:13, layers/nasfpn.py:289)"""
    if nn.dry_run():
        return _dry(x.shape, x)
    return _ActFn.apply(x, K.ACT_SWISH)


class _Relu6Fn(Function):
    @staticmethod
    def forward(ctx, x):
        xc = _c(x)
        ctx.save_for_backward(xc)
        return K.clip_fwd(xc, 0.0, 6.0)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        return K.clip_bwd(xc, _c(dy), 0.0, 6.0)


def relu6(x):
    """tf.nn.relu6 (backbones/mobilenetv2_common.py:50,61,161,168) = clip to [0, 6]; the gradient passes inside the interval (TF's Relu6Grad is
    strict at the two end points, the clip gradient is not: a difference on a set of measure zero)"""
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    return _Relu6Fn.apply(x)


class _AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.axpby(_c(a), _c(b), 1.0, 1.0)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    if nn.dry_run():
        return _dry(a.shape, a)
    return _AddFn.apply(a, b)


class _ForkFn(Function):
    """One tensor, several consumers.  torch.autograd would add the consumers' gradients with its own elementwise kernels; here every
    consumer gets an alias, and the backward node sums the arriving gradients with iseg_axpby -- no ATen arithmetic on the step."""

    @staticmethod
    def forward(ctx, x, n):
        # an alias nobody differentiates through must arrive as None, not as a tensor of zeros the engine fills with an ATen kernel and this node
        # then adds (Swin-T + FPN: two such maps of 100 MB and 25 MB per step)
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        live = [_c(g) for g in grads if g is not None]
        if not live:
            return None, None
        total = live[0]
        for i, g in enumerate(live[1:]):
            if g.dtype != total.dtype:
                g = K.cast(g, total.dtype)
            total = K.axpby(total, g, 1.0, 1.0, out=(total if i > 0 else None))      # the first sum allocates, the rest accumulate in place
        return total, None


def fork(x, n=2):
    """n aliases of x, one per consumer (see _ForkFn); a no-op outside autograd"""
    if nn.dry_run() or n <= 1 or not (torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled()):
        return (x,) * n
    return _ForkFn.apply(x, n)


_RNG_COUNTER = [0]


def next_seed():
    """one stream per (global seed, data-parallel rank, draw): replicas must not share dropout / drop-path masks on their shards"""
    _RNG_COUNTER[0] += 1
    return (nn.seed() * 0x9E3779B97F4A7C15 + _RNG_COUNTER[0] * 0xD1B54A32D192ED03 + dist.rank() * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF


class _DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, rate, seed):
        ctx.rate, ctx.seed = rate, seed
        return K.dropout(_c(x), rate, seed)

    @staticmethod
    def backward(ctx, dy):
        return K.dropout(_c(dy), ctx.rate, ctx.seed), None, None


def dropout(x, rate, training):
    if not training or rate <= 0:
        return x
    return _DropoutFn.apply(x, float(rate), next_seed())


class _RowScaleFn(Function):
    @staticmethod
    def forward(ctx, x, s):
        C = x.shape[-1]
        rpg = x.numel() // C // x.shape[0]
        ctx.rpg = rpg
        ctx.save_for_backward(s)
        return K.rowscale(_c(x).reshape(-1, C), s, rpg).reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (s,) = ctx.saved_tensors
        C = dy.shape[-1]
        return K.rowscale(_c(dy).reshape(-1, C), s, ctx.rpg).reshape(dy.shape), None


class _DropPathPool:
    """The drop-path call sites of a training step draw from ONE launch: the first step with a given sample count records the
    (samples, keep probability) sequence, every later step with that sample count replays it -- one [P, n] tensor of factors, rows handed
    out in call order.  Plans are kept PER SAMPLE COUNT (round 4): a smaller last batch, or a second crop size of a graphed trainer, no longer
    throws the plan away -- which had made the step after every switch draw one seed per call site instead of one per step, i.e. made the
    random stream depend on the order of batch shapes (and a replayed graph, which froze the pooled form, disagree with the eager run).  A step
    whose sequence differs from its plan (eval in between, a changed model) falls back to one launch per call and re-records."""

    def __init__(self):
        self.active, self.plans, self.rec, self.masks, self.cursor, self.keeps, self.plan = False, {}, None, None, 0, {}, None

    def begin(self):
        if self.rec:      # file the previous step's sequence under its sample count
            counts = {q[0] for q in self.rec}
            if len(counts) == 1:
                n = next(iter(counts))
                if self.plans.get(n) != self.rec:
                    self.plans[n] = list(self.rec)
                    self.keeps.pop(n, None)
        self.active, self.rec, self.cursor, self.masks, self.plan = True, [], 0, None, None

    def end(self):
        self.active = False

    def take(self, n, keep, device):
        keep = float(keep)
        if not self.active:
            return K.drop_path_mask(n, keep, next_seed(), device)
        self.rec.append((n, keep))
        k = self.cursor
        if k == 0 and self.masks is None and len(self.rec) == 1:
            self.plan = self.plans.get(n)
        plan = self.plan
        if plan is not None and k < len(plan) and plan[k] == (n, keep):
            if self.masks is None:
                keeps = self.keeps.get(n)
                if keeps is None or keeps.device != device:
                    keeps = self.keeps[n] = torch.tensor([q[1] for q in plan], dtype=torch.float32, device=device)
                self.masks = K.drop_path_masks(keeps, n, next_seed())
            self.cursor = k + 1
            return self.masks[k]
        self.plan = None      # the sequence differs from the plan: individual launches until the next step has re-recorded it
        return K.drop_path_mask(n, keep, next_seed(), device)


_DROP_PATH_POOL = _DropPathPool()


class drop_path_pool:
    """with F.drop_path_pool(): one training step's forward (CoreTrain's step uses it)"""

    def __enter__(self):
        _DROP_PATH_POOL.begin()

    def __exit__(self, *a):
        _DROP_PATH_POOL.end()


def drop_path_factors(n, keep_prob, device):
    """per-sample factors floor(keep + u) / keep (utils/drops.py:14-20) for one call site"""
    return _DROP_PATH_POOL.take(int(n), keep_prob, device)


def drop_path(x, drop_prob, training, mask=None):
    """utils/drops.py:8-22.  `mask` (per-sample factors floor(keep+u)/keep) can be injected for parity tests."""
    if (not training) or drop_prob == 0.0:
        return x
    if mask is None:
        mask = drop_path_factors(x.shape[0], 1.0 - drop_prob, x.device)
    return _RowScaleFn.apply(x, mask)


# ---------------------------------------------------------------------------------------------------------
# resize, pooling, broadcast, concat
# ---------------------------------------------------------------------------------------------------------
class _ResizeBilinearFn(Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, out_dtype, align_corners=False):
        ctx.in_shape, ctx.in_dtype, ctx.align = x.shape, x.dtype, bool(align_corners)
        return K.resize_bilinear(_c(x), Ho, Wo, out_dtype=out_dtype, align_corners=ctx.align)

    @staticmethod
    def backward(ctx, dy):
        _, Hi, Wi, _ = ctx.in_shape
        return K.resize_bilinear_bwd(_c(dy), Hi, Wi, ctx.in_dtype, align_corners=ctx.align), None, None, None, None


def resize_bilinear(x, size, out_dtype=None, align_corners=False):
    """tf.image.resize(method="bilinear") (half-pixel centres); align_corners=True: tf.compat.v1.image.resize(..., align_corners=True)"""
    Ho, Wo = int(size[0]), int(size[1])
    if nn.dry_run():
        return _dry((x.shape[0], Ho, Wo, x.shape[3]), x, out_dtype)
    if x.shape[1] == Ho and x.shape[2] == Wo and (out_dtype is None or out_dtype == x.dtype):
        return x
    return _ResizeBilinearFn.apply(x, Ho, Wo, out_dtype or x.dtype, bool(align_corners))


class _GlobalAvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        N, H, W, C = x.shape
        xc = _c(x)
        out = torch.empty((N, C), dtype=torch.float32, device=x.device)
        K.colsum(xc, C, H * W * C, N, H * W, C, out, scale=1.0 / (H * W))
        ctx.shape = x.shape
        return K.cast(out, x.dtype).reshape(N, 1, 1, C)

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C = ctx.shape
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        K.broadcast_rows(_c(dy).reshape(N, C), dx, C, H * W * C, N, H * W, C, scale=1.0 / (H * W))
        return dx


def global_avg_pool(x):
    """tf.reduce_mean(x, axis=(1,2), keepdims=True)"""
    if nn.dry_run():
        return _dry((x.shape[0], 1, 1, x.shape[3]), x)
    return _GlobalAvgPoolFn.apply(x)


class _BroadcastHWFn(Function):
    @staticmethod
    def forward(ctx, v, H, W):
        N, _, _, C = v.shape
        y = torch.empty((N, H, W, C), dtype=v.dtype, device=v.device)
        K.broadcast_rows(_c(v).reshape(N, C), y, C, H * W * C, N, H * W, C)
        ctx.hw = (H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C = dy.shape
        out = torch.empty((N, C), dtype=torch.float32, device=dy.device)
        dy2, lddy = _rows2d(dy)
        K.colsum(dy2, lddy, H * W * lddy, N, H * W, C, out)
        return K.cast(out, dy.dtype).reshape(N, 1, 1, C), None, None


def broadcast_hw(v, H, W):
    """tf.ones([1,H,W,1]) * v for v [N,1,1,C] (layers/model_builder.py:268)"""
    if nn.dry_run():
        return _dry((v.shape[0], H, W, v.shape[3]), v)
    return _BroadcastHWFn.apply(v, int(H), int(W))


class _ConcatFn(Function):
    @staticmethod
    def forward(ctx, *xs):
        lead = xs[0].shape[:-1]
        widths = [x.shape[-1] for x in xs]
        total = sum(widths)
        rows = xs[0].numel() // widths[0]
        y = torch.empty((*lead, total), dtype=xs[0].dtype, device=xs[0].device)
        y2 = y.reshape(rows, total)
        off = 0
        for x, wd in zip(xs, widths):
            K.copy2d(_c(x).reshape(rows, wd), wd, y2[:, off:off + wd], total, rows, wd)
            off += wd
        ctx.widths = widths
        return y

    @staticmethod
    def backward(ctx, dy):
        outs, off = [], 0
        for wd in ctx.widths:
            outs.append(dy[..., off:off + wd])   # strided views; consumers make them contiguous only if they must
            off += wd
        return tuple(outs)


def concat(xs):
    """tf.concat(xs, axis=-1)"""
    if nn.dry_run():
        return _dry((*xs[0].shape[:-1], sum(x.shape[-1] for x in xs)), xs[0])
    return _ConcatFn.apply(*xs)


# ---------------------------------------------------------------------------------------------------------
# ConvNeXt block, fused:  x + drop_path(gamma * pw2(gelu(pw1(LN(dw7x7(x))))))     backbones/convnext.py:47-63
# ---------------------------------------------------------------------------------------------------------
_BATCHED_PREP = os.environ.get("ISEG_BATCHED_PREP", "1") == "1"      # 0: per-block prep launches (A/B measurements)
_MLP_LN_ON_LOAD = os.environ.get("ISEG_MLP_LN_ON_LOAD", "1") == "1"      # 0: LayerNorm of the fused stages as its own kernel (A/B measurements)
_WGRAD_PAIR = os.environ.get("ISEG_WGRAD_PAIR", "1") == "1"      # 0: the two weight-gradient products of an un-fused block as two launches (A/B measurements)
_LAYERSCALE_FROM_SLABS = os.environ.get("ISEG_LAYERSCALE_FROM_SLABS", "1") == "1"      # 0: slab sum + Z tensor + layer-scale kernel (A/B measurements)
_MLP_LN_BWD_FUSED = os.environ.get("ISEG_MLP_LN_BWD_FUSED", "1") == "1"      # 0: LayerNorm backward of the fused stages as its own kernel (A/B measurements)
# 1: the drop-path factor rides the saved activation of the un-fused stages and rowscale_kernel disappears (round 6, see _ConvNeXtBlockFn.forward).
# Measured on the flagship, interleaved on one box: 8.196 / 8.210 / 8.227 ms without, 8.225 / 8.237 / 8.256 ms with it -- the 12 row-scale passes it
# removes (25 MB each) come back as the row-sum job's read of dout, two wider epilogues and wider partial rows.  Off by default; kept for the
# launch count (11 instead of 12 per stage-2 block) and tested (tests/test_blocks_gpu.py, tests/test_kernels_gpu.py).
_DP_FOLDED = os.environ.get("ISEG_DP_FOLDED", "0") == "1"
_SLAB_REDUCE_MERGED = os.environ.get("ISEG_SLAB_REDUCE_MERGED", "1") == "1"      # 0: slab sum of dW1 as its own launch (A/B measurements)
_MLP_BWD_NO_HIDDEN = os.environ.get("ISEG_MLP_BWD_NO_HIDDEN", "1") == "1"      # 0: the round-2 backward route of the fused stages (A/B measurements)


class _ConvNeXtBlockFn(Function):
    @staticmethod
    def forward(ctx, x, dw_kernel, dw_bias, ln_gamma, ln_beta, w1, b1, w2, b2, gamma, dil, eps, dp_mask):
        import types

        p = types.SimpleNamespace(dw_kernel=dw_kernel, dw_bias=dw_bias, ln_gamma=ln_gamma, ln_beta=ln_beta, w1=w1, b1=b1, w2=w2,
                                  b2=b2, gamma=gamma)
        N, H, W, C = x.shape
        xc = _c(x)
        Kk = p.dw_kernel.shape[0]
        pad = (Kk - 1) * dil // 2
        y1 = K.dwconv2d(xc, p.dw_kernel.data.reshape(Kk * Kk, C), p.dw_bias.data, Kk, dil, pad, pad)
        M = N * H * W
        ctx.fused = K.convnext_mlp_supported(C, xc.dtype)
        ctx.dp_folded = False
        # round 3: on the fused stages LayerNorm rides the MLP kernels' row loads (forward: statistics + normalisation, backward: normalisation
        # from the saved statistics), so y2 is never written or read -- needs the backward route that keeps nothing [M, 4C]-shaped either
        ctx.ln_on_load = ctx.fused and _MLP_LN_ON_LOAD and _MLP_BWD_NO_HIDDEN and (dp_mask is None or (H * W) % 64 == 0)
        if ctx.ln_on_load:
            y2 = mean = rstd = None
        else:
            y2, mean, rstd = K.layernorm_fwd(y1.reshape(M, C), p.ln_gamma.data, p.ln_beta.data, eps)
        grad = any(ctx.needs_input_grad)          # grad mode is off inside Function.forward; this is the tape's view
        # measured on MI355X (tools/kbench_gemm.py): writing both h and g = gelu(h) from the pw1 epilogue (175+83+117 us at
        # stage 0 for pw1/pw2/wgrad2) beats writing h only and re-deriving gelu(h) while staging the A operand of pw2 and of
        # the pw2 weight gradient (83+147+185 us): the transform sits on the load->LDS critical path.  a_act stays available.
        # h holds gelu'(pre-activation), not the pre-activation: the pw1 epilogue has Phi and exp(-v^2/2) in hand for gelu anyway, and
        # the backward epilogue then costs one multiply instead of an erf per element (+50 us of VALU per 100 M elements, measured)
        gam = p.gamma.data if p.gamma is not None else None
        if ctx.fused:
            # wide stages (C = 96 / 192, bf16): the [M, 4C] hidden tile stays on the CU (csrc/mlp_fused.hip) and the backward pass
            # recomputes it, so nothing [M, 4C]-shaped is kept; `bw` holds the tiled weight images the backward chain streams
            if _BATCHED_PREP:      # one launch per weight update for every block of the model (nn.mlp_tiled)
                fw, bw = nn.mlp_tiled(p.w1, p.w2, p.gamma)
            else:
                fw, bw = K.convnext_mlp_prep(p.w1.data, p.w2.data, gam, backward=grad)
            if ctx.ln_on_load:
                out, mean, rstd = K.convnext_mlp_fwd_ln(y1.reshape(M, C), p.ln_gamma.data, p.ln_beta.data, eps, fw, p.b1.data, p.b2.data, gam,
                                                        dp_mask, H * W, xc.reshape(M, C))
            else:
                out = K.convnext_mlp_fwd(y2, fw, p.b1.data, p.b2.data, gam, dp_mask, H * W, xc.reshape(M, C))
            h, g = bw, None
        else:
            h = torch.empty((M, 4 * C), dtype=xc.dtype, device=xc.device) if grad else None
            w1t, w2t = (nn.wt(p.w1), nn.wt(p.w2)) if xc.dtype == torch.bfloat16 else (None, None)
            # round 6: the drop-path factor s rides the SAVED activation.  x + s gamma (g W2 + b2) = x + gamma ((s g) W2 + s b2): the pwconv1 epilogue
            # writes s g (gelu' stays unscaled), pwconv2 scales its bias row-wise instead of its result, and the backward pass then works on the
            # UNSCALED dout -- the row factor goes into the x-aux epilogue of dH, Z = (s g)^T dout needs no scaled operand, and the column sums
            # S = colsum(s dout) are formed inside the layer-scale launch: rowscale_kernel (one pass over [M, C] per block) disappears.
            ctx.dp_folded = bool(_DP_FOLDED and grad and dp_mask is not None and w1t is not None and w2t is not None and p.gamma is not None and
                                 p.b1 is not None and _WGRAD_PAIR and _LAYERSCALE_FROM_SLABS and 4 * C >= 768 and C % 8 == 0)
            if ctx.dp_folded:
                g = K.dense_fwd_t(y2, w1t, p.b1.data, act=K.ACT_GELU, pre_out=h, pre_deriv=True, rowscale=dp_mask, rows_per_group=H * W)
                out = K.dense_fwd_t(g, w2t, p.b2.data, colscale=gam, rowscale=dp_mask, rows_per_group=H * W, residual=xc.reshape(M, C),
                                    bias_rowscaled=True)
            elif w1t is not None and w2t is not None:
                # K-contiguous kernel copies: the forward products run on the LDS-DMA GEMM like the data gradients (256 x 128 tiles)
                g = K.dense_fwd_t(y2, w1t, p.b1.data, act=K.ACT_GELU, pre_out=h, pre_deriv=grad)
                out = K.dense_fwd_t(g, w2t, p.b2.data, colscale=gam, rowscale=dp_mask, rows_per_group=H * W, residual=xc.reshape(M, C))
            else:
                g = K.dense_fwd(y2, nn.w(p.w1), p.b1.data, act=K.ACT_GELU, pre_out=h, pre_deriv=grad)
                out = K.dense_fwd(g, nn.w(p.w2), p.b2.data, colscale=gam, rowscale=dp_mask, rows_per_group=H * W, residual=xc.reshape(M, C))
        ctx.p, ctx.dil, ctx.pad = p, dil, pad
        ctx.save_for_backward(xc, y1, y2, mean, rstd, h, g if grad else None, dp_mask)
        return out.reshape(N, H, W, C)

    @staticmethod
    def backward(ctx, dout):
        """Two queues.  The data-gradient chain (dbr -> dh -> dy2 -> LayerNorm -> depthwise data gradient) is what the next block waits
        for; everything that only feeds the optimizer (column sums, the two weight-gradient GEMMs, layer-scale gradients, the depthwise
        weight gradient) goes to a side HIP stream behind events, so the short kernels of the narrow stages overlap instead of queueing
        (_SideQueue; opt-in with ISEG_SIDE_STREAM=1 -- measured slower than one queue on one GPU, see _side_enabled).  Both queues meet before the gradients are announced to the reducer."""
        xc, y1, y2, mean, rstd, h, g, dp_mask = ctx.saved_tensors
        p, dil, pad = ctx.p, ctx.dil, ctx.pad
        N, H, W, C = xc.shape
        M = N * H * W
        Kk = p.dw_kernel.shape[0]
        do2 = _c(dout).reshape(M, C)
        cdt = xc.dtype
        side = _SideQueue(xc.device)
        if ctx.fused and _MLP_BWD_NO_HIDDEN and (dp_mask is None or (H * W) % 64 == 0):
            # round 3: nothing [M, 4C]-shaped reaches HBM in the backward pass either.  One kernel carries the data gradient through the
            # recomputed hidden tile; a second one (workgroups own 128 hidden units and a chunk of rows) recomputes it again and contracts
            # over the rows for every parameter gradient of the MLP; the drop-path row factor and the column sums of dbr ride both
            # (csrc/mlp_wgrad.hip) -- replaces rowscale + colsum + chain + two weight-gradient GEMMs + their split-K sums + layerscale_grads
            bw = h
            yop, ln = (y1.reshape(M, C), (mean, rstd, p.ln_gamma.data, p.ln_beta.data)) if ctx.ln_on_load else (y2, None)
            dy1 = dy2 = None
            if ln is not None and _MLP_LN_BWD_FUSED:      # the chain kernel's epilogue carries the rows through the LayerNorm backward too
                dy1 = K.convnext_mlp_bwd_data_ln(yop, do2, bw, p.b1.data, ln, _grad(p.ln_gamma), _grad(p.ln_beta), dp_mask, H * W)
            else:
                dy2 = K.convnext_mlp_bwd_data(yop, do2, bw, p.b1.data, dp_mask, H * W, ln=ln)
            side.run(lambda: K.convnext_mlp_wgrad(yop, do2, bw, p.b1.data, p.w2.data, p.b2.data, p.gamma.data if p.gamma is not None else None,
                                                  _grad(p.w1), _grad(p.b1), _grad(p.w2), _grad(p.b2),
                                                  _grad(p.gamma) if p.gamma is not None else None, dp_mask, H * W, ln=ln), yop, do2)
            del h, bw
        else:
            dy1 = None
            dy2 = _ConvNeXtBlockFn._mlp_backward_with_hidden(ctx, p, do2, y2, h, g, dp_mask, side, H, W, C, M, cdt, xc)
            del h, g
        if dy1 is None:
            dy1 = K.layernorm_bwd(dy2, y1.reshape(M, C), p.ln_gamma.data, mean, rstd, _grad(p.ln_gamma), _grad(p.ln_beta))
        dy1 = dy1.reshape(N, H, W, C)
        side.run(lambda: K.dwconv2d_bwd_weight(xc, dy1, _grad(p.dw_kernel).reshape(Kk * Kk, C), _grad(p.dw_bias), Kk, dil, pad, pad), xc, dy1)
        dx = None
        if ctx.needs_input_grad[0]:
            padb = (Kk - 1) * dil - pad
            dx = K.dwconv2d(dy1, p.dw_kernel.data.reshape(Kk * Kk, C), None, Kk, dil, padb, padb, flip=True, add=_c(dout))
        side.join()
        dist.grads_ready(p.dw_kernel, p.dw_bias, p.ln_gamma, p.ln_beta, p.w1, p.b1, p.w2, p.b2, p.gamma)
        return (dx,) + (None,) * 12

    @staticmethod
    def _mlp_backward_with_hidden(ctx, p, do2, y2, h, g, dp_mask, side, H, W, C, M, cdt, xc):
        """the round-2 route: g / dh materialised ([M, 4C] each), two weight-gradient GEMMs; still the path of the un-fused stages (C >= 384)"""
        if ctx.dp_folded:      # (forward: g = s gelu(h) was saved; see there)
            w2eff = nn.w_colscaled(p.w2, p.gamma) if _BATCHED_PREP else K.scale_cols_cast(p.w2.data, p.gamma.data, cdt)
            dh = K.dense_dgrad(do2, w2eff, act=K.ACT_MUL_AUX, aux=h, rowscale=dp_mask, rows_per_group=H * W)      # s (dout W2g^T) gelu'(pre)
            del h

            def folded_param_grads():
                srow = (do2, dp_mask, H * W)
                sl = K.dense_wgrad_pair(g, do2, y2, dh, _grad(p.w1), _grad(p.b1), defer_second=_SLAB_REDUCE_MERGED, ones_first=False)
                if sl is not None:      # stage 2: both weight gradients in one launch, every slab sum + S in the layer-scale launch
                    K.layerscale_grads_slabs(sl[0], sl[1], p.w2.data, p.b2.data, p.gamma.data, _grad(p.w2), _grad(p.gamma), _grad(p.b2),
                                             extra=sl[2] if len(sl) > 2 else None, srow=srow)
                    return
                slz = K.dense_wgrad_slabs(g, do2, ones_row=False)      # stage 3 (the two products do not pair): Z = (s g)^T dout, no ones-row
                if slz is not None:
                    K.layerscale_grads_slabs(slz[0], slz[1], p.w2.data, p.b2.data, p.gamma.data, _grad(p.w2), _grad(p.gamma), _grad(p.b2), srow=srow)
                else:      # (not split: the tensor form, with S from a scaled copy after all)
                    Z = torch.empty((4 * C, C), dtype=torch.float32, device=xc.device)
                    K.dense_wgrad(g, do2, Z, accumulate=False)
                    S_ = torch.empty(C, dtype=torch.float32, device=xc.device)
                    K.colsum(K.rowscale(do2, dp_mask, H * W), C, 0, 1, M, C, S_)
                    K.layerscale_grads(Z, p.w2.data, p.b2.data, p.gamma.data, S_, _grad(p.w2), _grad(p.gamma), _grad(p.b2))
                K.dense_wgrad(y2, dh, _grad(p.w1), bias_grad=_grad(p.b1))

            side.run(folded_param_grads, do2, g, dh, y2)
            return K.dense_dgrad(dh, nn.w(p.w1))
        dbr = K.rowscale(do2, dp_mask, H * W) if dp_mask is not None else do2
        S = torch.empty(C, dtype=torch.float32, device=xc.device)      # column sums of dbr (layer-scale and bias gradients)
        s_on_gemm = p.gamma is not None and (ctx.fused or g is not None) and xc.dtype == torch.bfloat16 and 4 * C >= 768      # rides Z = g^T dbr below
        if not s_on_gemm:
            K.colsum(dbr, C, 0, 1, M, C, S)
        dy2 = None
        if ctx.fused:
            # h is the tiled weight buffer here: g = gelu(pre), dh = (dbr @ (W2 gamma)^T) * gelu'(pre), dy2 = dh @ W1^T in one launch
            g, dh, dy2 = K.convnext_mlp_bwd(y2, dbr, h, p.b1.data)
        else:
            if p.gamma is None:
                w2eff = nn.w(p.w2)
            elif _BATCHED_PREP and cdt == torch.bfloat16:
                w2eff = nn.w_colscaled(p.w2, p.gamma)
            else:
                w2eff = K.scale_cols_cast(p.w2.data, p.gamma.data, cdt)
            dh = K.dense_dgrad(dbr, w2eff, act=K.ACT_MUL_AUX, aux=h)          # [M,4C] = (dbr @ W2g^T) * gelu'(pre), h = gelu'(pre)
        del h

        # --- side: pw2 + layer scale from Z = g^T dbr and S = colsum(dbr) (no pass over [M,C] for gamma), then dW1 (+ db1)
        def param_grads():
            if p.gamma is not None:
                if s_on_gemm and _LAYERSCALE_FROM_SLABS and _WGRAD_PAIR and p.b1 is not None:
                    # round 5: both weight-gradient products of the block as ONE launch (36 tiles x 7 splits instead of 2 x 18 x 13): Z stops at its
                    # slabs for the layer-scale kernel, dW1 / db1 are summed by the generic slab reduce
                    # (round 6: the slab sum of dW1 / db1 rides the layer-scale launch -- _SLAB_REDUCE_MERGED=0 keeps the two launches)
                    sl = K.dense_wgrad_pair(g, dbr, y2, dh, _grad(p.w1), _grad(p.b1), defer_second=_SLAB_REDUCE_MERGED)
                    if sl is not None:
                        K.layerscale_grads_slabs(sl[0], sl[1], p.w2.data, p.b2.data, p.gamma.data, _grad(p.w2), _grad(p.gamma), _grad(p.b2),
                                                 extra=sl[2] if len(sl) > 2 else None)
                        return
                sl = K.dense_wgrad_slabs(g, dbr) if (s_on_gemm and _LAYERSCALE_FROM_SLABS) else None
                if sl is not None:      # the layer-scale kernel sums the split-K slabs of Z (and its ones-row S) while it reads them
                    K.layerscale_grads_slabs(sl[0], sl[1], p.w2.data, p.b2.data, p.gamma.data, _grad(p.w2), _grad(p.gamma), _grad(p.b2))
                else:
                    Z = torch.empty((4 * C, C), dtype=torch.float32, device=xc.device)
                    K.dense_wgrad(g, dbr, Z, accumulate=False, bias_grad=S if s_on_gemm else None)      # Z = gelu(h)^T dbr (+ S from its ones-row)
                    K.layerscale_grads(Z, p.w2.data, p.b2.data, p.gamma.data, S, _grad(p.w2), _grad(p.gamma), _grad(p.b2))
            else:
                K.dense_wgrad(g, dbr, _grad(p.w2))
                K.axpby(S, _grad(p.b2), 1.0, 1.0, out=_grad(p.b2))
            K.dense_wgrad(y2, dh, _grad(p.w1), bias_grad=_grad(p.b1))      # db1 rides the wgrad GEMM (virtual ones-row) when C % 128 != 0

        side.run(param_grads, dbr, g, dh, y2)
        if not ctx.fused:
            dy2 = K.dense_dgrad(dh, nn.w(p.w1))                                # [M,C]
        return dy2


_GRN_FOLD_RATIO = int(os.environ.get("ISEG_V2_GRN_FOLD_RATIO", "1"))      # experiments: fold from HW * ratio >= 2 * (4C) on


class _ConvNeXtV2BlockFn(Function):
    """One tape node for a ConvNeXt V2 block (backbones/convnext_v2.py:83-98): depthwise 7x7 -> LayerNorm -> Dense 4C -> GELU -> GRN ->
    Dense C -> drop path -> + inputs.  The first product's epilogue writes gelu(h) and gelu'(h); the second product's epilogue applies the
    drop-path row factor and adds the block input; in the backward pass the GELU derivative rides the GRN data-gradient kernel, the bias
    gradients ride the weight-gradient GEMMs and the residual gradient rides the depthwise data-gradient kernel -- no elementwise pass of
    its own for any of them."""

    @staticmethod
    def forward(ctx, x, dw_kernel, dw_bias, ln_gamma, ln_beta, w1, b1, grn_gamma, grn_beta, w2, b2, dil, eps, grn_eps, dp_mask):
        import types

        p = types.SimpleNamespace(dw_kernel=dw_kernel, dw_bias=dw_bias, ln_gamma=ln_gamma, ln_beta=ln_beta, w1=w1, b1=b1, w2=w2, b2=b2,
                                  grn_gamma=grn_gamma, grn_beta=grn_beta)
        N, H, W, C = x.shape
        xc = _c(x)
        M = N * H * W
        Kk = dw_kernel.shape[0]
        pad = (Kk - 1) * dil // 2
        y1 = K.dwconv2d(xc, dw_kernel.data.reshape(Kk * Kk, C), dw_bias.data, Kk, dil, pad, pad)
        y2, mean, rstd = K.layernorm_fwd(y1.reshape(M, C), ln_gamma.data, ln_beta.data, eps)
        grad = any(ctx.needs_input_grad)
        d = torch.empty((M, 4 * C), dtype=xc.dtype, device=xc.device) if grad else None      # gelu'(pre-activation)
        w1t, w2t = (nn.wt(w1), nn.wt(w2)) if xc.dtype == torch.bfloat16 else (None, None)
        if w1t is not None and w2t is not None:
            g = K.dense_fwd_t(y2, w1t, b1.data, act=K.ACT_GELU, pre_out=d, pre_deriv=grad)
        else:
            g = K.dense_fwd(y2, nn.w(w1), b1.data, act=K.ACT_GELU, pre_out=d, pre_deriv=grad)
        # Wide planes (at least two activation rows per kernel row and sample): the normalisation is folded into the second product --
        # per-sample kernels diag(gamma*nx_n + 1) W2 and the bias b2 + beta W2 -- so grn(g) is never written (csrc/grn.hip, "folded")
        HW = H * W
        ctx.fold = (w2t is not None and HW % 256 == 0 and HW * _GRN_FOLD_RATIO >= 8 * C and C >= 64 and C % 16 == 0      # (the LDS-DMA GEMM's own conditions)
                    and os.environ.get("ISEG_V2_GRN_FOLD", "1") == "1")
        if ctx.fold:
            nx, gx = K.grn_stats(g.reshape(N, HW, 4 * C), grn_eps)
            w2n = K.grn_fold_weights(w2t, grn_gamma.data.reshape(-1), nx)
            bias2 = K.grn_fold_bias(w2.data.reshape(4 * C, C), grn_beta.data.reshape(-1), b2.data)
            out = torch.empty((M, C), dtype=xc.dtype, device=xc.device)
            K.gemm(g, w2n, out, M, C, 4 * C, lda=4 * C, ldb=4 * C, ldd=C, a_kcontig=1, b_kcontig=1, bias=bias2, rowscale=dp_mask,
                   rows_per_group=HW, residual=xc.reshape(M, C), ldr=C, b_group=(HW, 4 * C * C))
            z = None
        else:
            z, nx, gx = K.grn_fwd(g.reshape(N, HW, 4 * C), grn_gamma.data, grn_beta.data, grn_eps)
            z = z.reshape(M, 4 * C)
            if w1t is not None and w2t is not None:
                out = K.dense_fwd_t(z, w2t, b2.data, rowscale=dp_mask, rows_per_group=HW, residual=xc.reshape(M, C))
            else:
                out = K.dense_fwd(z, nn.w(w2), b2.data, rowscale=dp_mask, rows_per_group=HW, residual=xc.reshape(M, C))
        ctx.p, ctx.dil, ctx.pad, ctx.grn_eps = p, dil, pad, grn_eps
        if grad:
            ctx.save_for_backward(xc, y1, y2, mean, rstd, d, g, z, nx, gx, dp_mask)
        return out.reshape(N, H, W, C)

    @staticmethod
    def backward(ctx, dout):
        xc, y1, y2, mean, rstd, d, g, z, nx, gx, dp_mask = ctx.saved_tensors
        p, dil, pad = ctx.p, ctx.dil, ctx.pad
        N, H, W, C = xc.shape
        M = N * H * W
        Kk = p.dw_kernel.shape[0]
        do2 = _c(dout).reshape(M, C)
        dbr = K.rowscale(do2, dp_mask, H * W) if dp_mask is not None else do2
        HW = H * W
        if ctx.fold:
            # G = g^T dbr per sample (per row chunk where a sample alone would not fill the CUs): dW2, the GRN statistics and dbeta all come
            # from these small products -- no pass over dz for them, and no grn(g) to read
            S = K.colsum(dbr, C, 0, 1, M, C, torch.empty(C, dtype=torch.float32, device=xc.device))
            K.axpby(S, _grad(p.b2), 1.0, 1.0, out=_grad(p.b2))
            tiles = -(-4 * C // 128) * -(-C // 128)
            sps = 1
            while N * sps * tiles < 384 and HW % (2 * sps) == 0 and HW // (2 * sps) >= 512:
                sps *= 2
            rows = HW // sps
            slabs = torch.empty((N * sps, 4 * C, C), dtype=torch.float32, device=xc.device)
            K.gemm(g, dbr, slabs, 4 * C, C, rows, lda=4 * C, ldb=C, ldd=C, a_kcontig=0, b_kcontig=0, batch=N * sps, batch_inner=1,
                   sa=(rows * 4 * C, 0), sb=(rows * C, 0), sd=(4 * C * C, 0))
            dstats = K.grn_fold_wgrad(slabs, sps, p.w2.data.reshape(4 * C, C), p.grn_gamma.data.reshape(-1), p.grn_beta.data.reshape(-1), nx, S,
                                      _grad(p.w2).reshape(4 * C, C))
            del slabs
            dz = K.dense_dgrad(dbr, nn.w(p.w2))                                       # [M, 4C]
            dh = K.grn_bwd_folded(dz.reshape(N, HW, 4 * C), g.reshape(N, HW, 4 * C), p.grn_gamma.data.reshape(-1), nx, gx, dstats,
                                  _grad(p.grn_gamma).reshape(-1), _grad(p.grn_beta).reshape(-1), ctx.grn_eps, mul=d).reshape(M, 4 * C)
        else:
            K.dense_wgrad(z, dbr, _grad(p.w2), bias_grad=_grad(p.b2))
            dz = K.dense_dgrad(dbr, nn.w(p.w2))                                       # [M, 4C]
            del z
            # GRN data gradient times gelu'(h) in one kernel; dgamma | dbeta from the same pass over dz
            dh = K.grn_bwd(dz.reshape(N, HW, 4 * C), g.reshape(N, HW, 4 * C), p.grn_gamma.data, nx, gx, _grad(p.grn_gamma).reshape(-1),
                           _grad(p.grn_beta).reshape(-1), ctx.grn_eps, mul=d).reshape(M, 4 * C)
        del dz, g, d
        K.dense_wgrad(y2, dh, _grad(p.w1), bias_grad=_grad(p.b1))
        dy2 = K.dense_dgrad(dh, nn.w(p.w1))                                            # [M, C]
        del dh
        dy1 = K.layernorm_bwd(dy2, y1.reshape(M, C), p.ln_gamma.data, mean, rstd, _grad(p.ln_gamma), _grad(p.ln_beta))
        dy1 = dy1.reshape(N, H, W, C)
        K.dwconv2d_bwd_weight(xc, dy1, _grad(p.dw_kernel).reshape(Kk * Kk, C), _grad(p.dw_bias), Kk, dil, pad, pad)
        dx = None
        if ctx.needs_input_grad[0]:
            padb = (Kk - 1) * dil - pad
            dx = K.dwconv2d(dy1, p.dw_kernel.data.reshape(Kk * Kk, C), None, Kk, dil, padb, padb, flip=True, add=_c(dout))
        dist.grads_ready(p.dw_kernel, p.dw_bias, p.ln_gamma, p.ln_beta, p.w1, p.b1, p.grn_gamma, p.grn_beta, p.w2, p.b2)
        return (dx,) + (None,) * 14


def convnext_v2_block(x, params, dilation, eps, grn_eps, dp_mask):
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    p = params
    return _ConvNeXtV2BlockFn.apply(x, p.dw_kernel, p.dw_bias, p.ln_gamma, p.ln_beta, p.w1, p.b1, p.grn_gamma, p.grn_beta, p.w2, p.b2,
                                    int(dilation), float(eps), float(grn_eps), dp_mask)


# ---------------------------------------------------------------------------------------------------------
# side queue for work that only feeds the optimizer
# ---------------------------------------------------------------------------------------------------------
_SIDE_STREAMS = {}


def _side_enabled():
    """independent work on extra HIP streams?  ISEG_SIDE_STREAM = 1 always, `capture` only while a HIP graph is being captured, 0 / unset never.
    Eager (flagship step, interleaved A/B): 9.08 ms with one queue, 9.39 ms with two -- every fork / join is a pair of host-side event calls.
    Inside a captured graph the forks are edges of the graph; measured replayed (one box, interleaved): flagship 9.12 / 9.16 ms with one queue,
    9.17 / 9.17 ms with two (an earlier box: 9.11 vs 8.98); ResNet-50 + ASPP 8.18 vs 11.6 ms, InternImage-B 36.6 vs 45.9 ms, Swin-T equal -- the
    runtime pays for every fork / join of a replayed graph, and the small-kernel models have hundreds.  Off by default."""
    import os

    mode = os.environ.get("ISEG_SIDE_STREAM", "0")
    if mode == "1":
        return True
    if mode != "capture" or not torch.cuda.is_available():
        return False
    return torch.cuda.is_current_stream_capturing()


_BRANCH_STREAMS = {}


def parallel_branches(fns, device=None):
    """[fn() for fn in fns], each on its own HIP stream when side streams are enabled (_side_enabled): the branches of ASPP are five
    independent conv -> BatchNorm -> ReLU chains of 64..256-workgroup kernels that leave most of the chip idle one at a time.  The autograd
    engine replays each branch's backward on the stream its forward ran on, so the backward pass forks the same way."""
    if len(fns) < 2 or not _side_enabled():
        return [fn() for fn in fns]
    main = torch.cuda.current_stream()
    key = main.device.index
    pool = _BRANCH_STREAMS.setdefault(key, [])
    while len(pool) < len(fns):
        pool.append(torch.cuda.Stream(device=main.device))
    outs = []
    for fn, st in zip(fns, pool):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs.append(fn())
    capturing = torch.cuda.is_current_stream_capturing()
    for o, st in zip(outs, pool):
        main.wait_stream(st)
        if not capturing:      # eager: tell the caching allocator that the main stream reads what a side stream allocated
            for t in (o if isinstance(o, (list, tuple)) else [o]):
                if torch.is_tensor(t):
                    t.record_stream(main)
    return outs


class _SideQueue:
    """run(fn, *tensors): enqueue fn on this device's side stream behind everything the current stream has enqueued so far; `tensors`
    are the buffers fn reads that the caller may release before the side stream gets to them (the caching allocator is told).
    join(): the current stream waits for the side stream.  Without ISEG_SIDE_STREAM=1 (or on the CPU) run() just calls fn."""

    def __init__(self, device):
        self.stream = None
        if device.type == "cuda" and _side_enabled():
            key = device.index if device.index is not None else torch.cuda.current_device()
            st = _SIDE_STREAMS.get(key)
            if st is None:
                st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
            self.stream = st
        self.used = False

    def run(self, fn, *tensors):
        if self.stream is None:
            return fn()
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            fn()
        for t in tensors:
            if t is not None:
                t.record_stream(self.stream)
        self.used = True

    def join(self):
        if self.stream is not None and self.used:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.used = False


def convnext_block(x, params, dilation, eps, dp_mask):
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    p = params
    return _ConvNeXtBlockFn.apply(x, p.dw_kernel, p.dw_bias, p.ln_gamma, p.ln_beta, p.w1, p.b1, p.w2, p.b2, p.gamma, int(dilation),
                                  float(eps), dp_mask)


# ---------------------------------------------------------------------------------------------------------
# ignore-label softmax cross-entropy          losses/catecrossentropy_ignore_label.py:44-88
# ---------------------------------------------------------------------------------------------------------
class _SoftmaxCEPerPixelFn(Function):
    """returns the per-position loss vector [N*H*W] exactly like the reference's weighted_loss (Reduction.NONE)"""

    @staticmethod
    def forward(ctx, logits, labels, num_class, ignore_label, class_w, focal=None):
        z = _c(logits).reshape(-1, num_class)
        if z.dtype != torch.float32:
            z = K.cast(z, torch.float32)
        y = _c(labels).reshape(-1)
        if y.dtype != torch.int32:
            y = y.to(torch.int32)
        px, _, _ = K.softmax_ce_ignore(z, y, ignore_label, class_w=class_w, want_px=True, focal=focal)
        ctx.args = (num_class, ignore_label, class_w, focal)
        ctx.in_dtype, ctx.in_shape = logits.dtype, logits.shape
        ctx.save_for_backward(z, y)
        return px

    @staticmethod
    def backward(ctx, dpx):
        z, y = ctx.saved_tensors
        num_class, ignore_label, class_w, focal = ctx.args
        gp = _c(dpx)
        if gp.dtype != torch.float32:
            gp = K.cast(gp, torch.float32)
        _, _, dz = K.softmax_ce_ignore(z, y, ignore_label, class_w=class_w, want_px=False, want_grad=True, grad_scale=1.0,
                                       grad_px=gp, focal=focal)
        if dz.dtype != ctx.in_dtype:
            dz = K.cast(dz, ctx.in_dtype)
        return dz.reshape(ctx.in_shape), None, None, None, None, None


class _SoftmaxCEMeanFn(Function):
    """mean over ALL positions (Keras' reduction of the NONE loss): loss and d(loss)/d(logits) in one fused pass"""

    @staticmethod
    def forward(ctx, logits, labels, num_class, ignore_label, class_w, weight, focal=None, cm=None):
        z = _c(logits).reshape(-1, num_class)
        if z.dtype != torch.float32:
            z = K.cast(z, torch.float32)
        y = _c(labels).reshape(-1)
        if y.dtype != torch.int32:
            y = y.to(torch.int32)
        P = z.shape[0]
        want_grad = ctx.needs_input_grad[0]
        _, s, dz = K.softmax_ce_ignore(z, y, ignore_label, class_w=class_w, want_px=False, want_sum=True, sum_scale=weight / P,
                                       want_grad=want_grad, grad_scale=weight / P, focal=focal, cm=cm)
        ctx.shape, ctx.in_dtype = logits.shape, logits.dtype
        ctx.save_for_backward(dz)
        return s.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (dz,) = ctx.saved_tensors
        # dloss is the scalar chain factor.  Inside CoreTrain's step it is exactly 1 (unit_loss_grad()), otherwise it is
        # applied on the device (a host read here would stall the stream)
        if not _UNIT_LOSS_GRAD[0]:
            dz = K.scale_dev(dz, _c(dloss).reshape(1).to(torch.float32))
        if dz.dtype != ctx.in_dtype:
            dz = K.cast(dz, ctx.in_dtype)
        return dz.reshape(ctx.shape), None, None, None, None, None, None, None


_UNIT_LOSS_GRAD = [False]


class unit_loss_grad:
    """context: the caller guarantees every scalar loss is differentiated with upstream gradient exactly 1"""

    def __enter__(self):
        self.prev = _UNIT_LOSS_GRAD[0]
        _UNIT_LOSS_GRAD[0] = True

    def __exit__(self, *a):
        _UNIT_LOSS_GRAD[0] = self.prev


def softmax_ce_per_pixel(logits, labels, num_class, ignore_label, class_w=None, focal=None):
    return _SoftmaxCEPerPixelFn.apply(logits, labels, num_class, ignore_label, class_w, focal)


# ---------------------------------------------------------------------------------------------------------
# Deferred logits upsample: inside CoreTrain's step the model hands the low-resolution logits to the loss, and one kernel does
# bilinear upsample + cross-entropy + its gradient through the resize + the confusion matrix  (layers/core_model_ext.py:199-256,
# losses/catecrossentropy_ignore_label.py:44-88, metrics/seg_metric_wrapper.py:89-102)
# ---------------------------------------------------------------------------------------------------------
_DEFER_UPSAMPLE = [False]


class defer_logits_upsample:
    """while active, SegManaged returns DeferredLogits instead of the bilinear-upsampled fp32 logits"""

    def __enter__(self):
        self.prev = _DEFER_UPSAMPLE[0]
        _DEFER_UPSAMPLE[0] = True

    def __exit__(self, *a):
        _DEFER_UPSAMPLE[0] = self.prev


def deferring_logits_upsample():
    return _DEFER_UPSAMPLE[0]


class DeferredLogits:
    """low-resolution logits [N,h,w,C] + the size tf.image.resize would take them to; quacks like the fp32 [N,H,W,C] tensor as far
    as shape checks go and turns into it on materialize()"""

    def __init__(self, low, size):
        self.low, self.size = low, (int(size[0]), int(size[1]))
        self.dtype = torch.float32

    @property
    def shape(self):
        return torch.Size((self.low.shape[0], self.size[0], self.size[1], self.low.shape[3]))

    def dim(self):
        return 4

    def fusable(self, num_class):
        N, h, w, Cc = self.low.shape
        return Cc == num_class and self.low.is_cuda and K.upsample_ce_supported(h, w, self.size[0], self.size[1], Cc)

    def materialize(self):
        return resize_bilinear(self.low, self.size, out_dtype=torch.float32)


class _UpsampleCEMeanFn(Function):
    @staticmethod
    def forward(ctx, z, labels, Ho, Wo, ignore_label, class_w, weight, cm):
        zc = _c(z)
        y = _c(labels)
        if y.dtype != torch.int32:
            y = y.to(torch.int32)
        P = zc.shape[0] * Ho * Wo
        s, dz = K.upsample_ce(zc, y, Ho, Wo, ignore_label, class_w=class_w, sum_scale=weight / P, want_grad=ctx.needs_input_grad[0],
                              grad_scale=weight / P, cm=cm)
        ctx.in_dtype = z.dtype
        ctx.save_for_backward(dz)
        return s.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (dz,) = ctx.saved_tensors
        if not _UNIT_LOSS_GRAD[0]:
            dz = K.scale_dev(dz, _c(dloss).reshape(1).to(torch.float32))
        if dz.dtype != ctx.in_dtype:
            dz = K.cast(dz, ctx.in_dtype)
        return dz, None, None, None, None, None, None, None


def upsample_softmax_ce_mean(deferred, labels, num_class, ignore_label, class_w=None, weight=1.0, cm=None):
    """mean over ALL positions of the ignore-label CE of the bilinear-upsampled logits; the upsampled tensor is never written"""
    return _UpsampleCEMeanFn.apply(deferred.low, labels, deferred.size[0], deferred.size[1], int(ignore_label), class_w, float(weight), cm)


def softmax_ce_mean(logits, labels, num_class, ignore_label, class_w=None, weight=1.0, focal=None, cm=None):
    """cm: int64 [C*C] confusion matrix that the same kernel pass updates with argmax(logits) (running mIoU of the train step)"""
    return _SoftmaxCEMeanFn.apply(logits, labels, num_class, ignore_label, class_w, float(weight), focal, cm)


# ---------------------------------------------------------------------------------------------------------
# replace_nan_or_inf (utils/op_utils.py:43-60), GroupNormalization, RMSNormalization, pooling
# ---------------------------------------------------------------------------------------------------------
class _ReplaceNanInfFn(Function):
    @staticmethod
    def forward(ctx, x, nan_value):
        xc = _c(x)
        ctx.save_for_backward(xc)
        return K.replace_nan_or_inf(xc, nan_value)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        return K.replace_nan_or_inf_bwd(xc, _c(dy)), None


def replace_nan_or_inf(x, nan_value=0.0):
    if nn.dry_run():
        return _dry(x.shape, x)
    return _ReplaceNanInfFn.apply(x, float(nan_value))


class _GroupNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps):
        N, C = x.shape[0], x.shape[-1]
        x3 = _c(x).reshape(N, -1, C)
        y, mean, rstd = K.groupnorm_fwd(x3, gamma.data if gamma is not None else None, beta.data if beta is not None else None, groups, eps)
        ctx.gamma, ctx.beta, ctx.groups = gamma, beta, groups
        ctx.save_for_backward(x3, mean, rstd)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x3, mean, rstd = ctx.saved_tensors
        g, b = ctx.gamma, ctx.beta
        dx = K.groupnorm_bwd(_c(dy).reshape(x3.shape), x3, g.data if g is not None else None, mean, rstd, ctx.groups,
                             _grad(g) if g is not None and g.requires_grad else None,
                             _grad(b) if b is not None and b.requires_grad else None)
        dist.grads_ready(g, b)
        return dx.reshape(dy.shape), None, None, None, None


def group_norm(x, gamma, beta, groups, eps):
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    return _GroupNormFn.apply(x, gamma, beta, int(groups), float(eps))


class _RMSNormFn(Function):
    @staticmethod
    def forward(ctx, x, scale, eps):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        y, rstd = K.rmsnorm_fwd(x2, scale.data, eps)
        ctx.scale = scale
        ctx.save_for_backward(x2, rstd)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, rstd = ctx.saved_tensors
        dx = K.rmsnorm_bwd(_c(dy).reshape(x2.shape), x2, ctx.scale.data, rstd, _grad(ctx.scale))
        dist.grads_ready(ctx.scale)
        return dx.reshape(dy.shape), None, None


def rms_norm(x, scale, eps):
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    return _RMSNormFn.apply(x, scale, float(eps))


class _GRNFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        N, C = x.shape[0], x.shape[-1]
        x3 = _c(x).reshape(N, -1, C)
        y, nx, gx = K.grn_fwd(x3, gamma.data, beta.data, eps)
        ctx.params = (gamma, beta)
        ctx.eps = eps
        ctx.save_for_backward(x3, nx, gx)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x3, nx, gx = ctx.saved_tensors
        gamma, beta = ctx.params
        dx = K.grn_bwd(_c(dy).reshape(x3.shape), x3, gamma.data, nx, gx, _grad(gamma), _grad(beta), ctx.eps)
        dist.grads_ready(gamma, beta)
        return dx.reshape(dy.shape), None, None, None


def grn(x, gamma, beta, eps=1e-6):
    """Global Response Normalization (backbones/convnext_v2.py:45-60): gamma * (x * nx) + beta + x with nx the per-sample channel response
    (L2 norm over H, W) divided by its mean over channels; gamma, beta are fp32 parameters of shape [1, 1, 1, C]"""
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(x.shape, x)
    return _GRNFn.apply(x, gamma, beta, float(eps))


class _Pool2dFn(Function):
    @staticmethod
    def forward(ctx, x, geom, mode):
        xc = _c(x)
        kh, kw, sh, sw, pt, pl, Ho, Wo = geom
        ctx.geom, ctx.mode = geom, mode
        ctx.save_for_backward(xc)
        return K.pool2d_fwd(xc, kh, kw, sh, sw, pt, pl, Ho, Wo, mode)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        kh, kw, sh, sw, pt, pl, _, _ = ctx.geom
        return K.pool2d_bwd(xc, _c(dy), kh, kw, sh, sw, pt, pl, ctx.mode), None, None


def _pool(x, pool_size, strides, padding, mode):
    kh, kw = (pool_size, pool_size) if isinstance(pool_size, int) else tuple(pool_size)
    if strides is None:
        strides = (kh, kw)
    sh, sw = (strides, strides) if isinstance(strides, int) else tuple(strides)
    Ho, Wo, pt, pl = _conv_geometry(x.shape[1], x.shape[2], kh, kw, (sh, sw), (1, 1),
                                    padding.lower() if isinstance(padding, str) else padding)
    if nn.dry_run():
        return _dry((x.shape[0], Ho, Wo, x.shape[3]), x)
    return _Pool2dFn.apply(x, (kh, kw, sh, sw, pt, pl, Ho, Wo), mode)


def max_pool2d(x, pool_size, strides=None, padding="same"):
    """keras.layers.MaxPooling2D / tf.nn.max_pool2d"""
    return _pool(x, pool_size, strides, padding, K.POOL_MAX)


def avg_pool2d(x, pool_size, strides=None, padding="same"):
    """tf.nn.avg_pool2d: padded cells do not count in the divisor"""
    return _pool(x, pool_size, strides, padding, K.POOL_AVG)


class _AddReluFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        y = K.add_relu(_c(a), _c(b))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        d = K.act_bwd(_c(dy), y, K.ACT_RELU)
        return d, d


def add_relu(a, b):
    """tf.nn.relu(tf.add(a, b))"""
    if nn.dry_run():
        return _dry(a.shape, a)
    if a.numel() % 8 != 0:
        return relu(add(a, b))
    return _AddReluFn.apply(a, b)


# ---------------------------------------------------------------------------------------------------------
# multi-head self-attention core on a packed [q | k | v] tensor:
#   softmax(scale * q k^T + bias[h] + mask[w]) (-> dropout) (-> clip) @ v
# backbones/swin.py:117-167 (bias, shift mask), keras MultiHeadAttention in backbones/vit.py:142-147, and
# layers/multihead_self_attention.py:106-150 (clip).  Score / context products are strided-batch GEMMs, one problem per
# (sample, head); probabilities are kept for the backward pass (288 GB HBM: cheaper than recomputing them).
# ---------------------------------------------------------------------------------------------------------
_MAX_GRID_Z = 65535


def _attn_chunks(B, heads):
    per = max(1, _MAX_GRID_Z // heads)
    return [(b0, min(B, b0 + per)) for b0 in range(0, B, per)]


class _FlashAttentionFn(Function):
    """global self-attention without bias / mask / clip / dropout, head_dim 64, bf16 (ViT, MHSA): online-softmax forward that keeps
    one log-sum-exp float per row, backward kernels that recompute the probabilities (csrc/flashattn.hip); no T x T tensor"""

    @staticmethod
    def forward(ctx, qkv, heads, scale):
        qkv = _c(qkv)
        out, lse = K.attention_fwd_train(qkv, heads, scale)
        ctx.cfg = (heads, scale)
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dO):
        qkv, out, lse = ctx.saved_tensors
        heads, scale = ctx.cfg
        return K.attention_bwd(qkv, out, _c(dO), lse, heads, scale), None, None


class _AttentionFn(Function):
    @staticmethod
    def forward(ctx, qkv, bias_table, heads, Cq, Cv, scale, bias_index, mask, windows, clip, drop_rate, seed, bias_window=0, want_probs=False):
        B, T, ld = qkv.shape
        assert ld == 2 * Cq + Cv
        qkv = _c(qkv)
        dq, dv = Cq // heads, Cv // heads
        ctx.fused = (bias_table is not None and Cq == Cv and clip is None and drop_rate <= 0 and not want_probs and
                     K.window_attention_supported(T, dq, qkv.dtype) and os.environ.get("ISEG_WINATTN", "1") != "0")
        if ctx.fused:      # Swin window attention: one wavefront per (window, head), probabilities never leave the CU
            bias = K.relpos_bias_gather(bias_table.data, bias_index, heads, T)
            ctx.cfg = (heads, Cq, Cv, scale, windows, clip, drop_rate, seed, 0)
            ctx.bias_table, ctx.bias_index, ctx.bias_window = bias_table, bias_index, bias_window
            table = K.window_attention_table(bias, mask, heads, T)
            ctx.save_for_backward(qkv, table)
            return K.window_attention_fwd(qkv, table, heads, scale)
        Tp = (T + 7) // 8 * 8
        dev, dtp = qkv.device, qkv.dtype
        P = torch.empty((B * heads, T, Tp), dtype=dtp, device=dev)
        O = torch.empty((B, T, Cv), dtype=dtp, device=dev)
        bias = None
        if bias_table is not None:
            bias = K.relpos_bias_gather(bias_table.data, bias_index, heads, T)
        qv, kv, vv = qkv[:, :, :Cq], qkv[:, :, Cq:2 * Cq], qkv[:, :, 2 * Cq:]
        Pd = P
        for b0, b1 in _attn_chunks(B, heads):
            nb = (b1 - b0) * heads
            Pz = P[b0 * heads:b1 * heads]
            K.gemm(qv[b0:b1], kv[b0:b1], Pz, T, T, dq, lda=ld, ldb=ld, ldd=Tp, a_kcontig=1, b_kcontig=1, alpha=scale, batch=nb,
                   batch_inner=heads, sa=(T * ld, dq), sb=(T * ld, dq), sd=(heads * T * Tp, T * Tp))
            # problems of a chunk start at a multiple of `windows` samples only if b0 % windows == 0: chunks are sample-aligned
            K.softmax_rows_fwd(Pz, nb, T, T, Tp, bias=bias, heads=heads, mask=mask, windows=windows)
        if drop_rate > 0:
            Pd = K.dropout(P, drop_rate, seed)
        ctx.pre_clip = None
        if clip is not None:      # the exact softmax output is kept for the backward pass; the clipped copy feeds P @ V
            ctx.pre_clip = Pd
            Pd = K.clip_fwd(Pd, clip[0], clip[1])
        for b0, b1 in _attn_chunks(B, heads):
            nb = (b1 - b0) * heads
            K.gemm(Pd[b0 * heads:b1 * heads], vv[b0:b1], O[b0:b1], T, dv, T, lda=Tp, ldb=ld, ldd=Cv, a_kcontig=1, b_kcontig=0,
                   batch=nb, batch_inner=heads, sa=(heads * T * Tp, T * Tp), sb=(T * ld, dv), sd=(T * Cv, dv))
        ctx.cfg = (heads, Cq, Cv, scale, windows, clip, drop_rate, seed, Tp)
        ctx.bias_table, ctx.bias_index, ctx.bias_window = bias_table, bias_index, bias_window
        ctx.save_for_backward(qkv, P, Pd if Pd is not P else None)
        if want_probs:      # the probabilities that multiply V (after dropout / clip), [B, heads, T, T]; no gradient flows through this output
            probs = Pd.reshape(B, heads, T, Tp)[..., :T]
            ctx.mark_non_differentiable(probs)
            return O, probs
        return O

    @staticmethod
    def backward(ctx, dO, *unused_dprobs):
        if ctx.fused:
            qkv, table = ctx.saved_tensors
            heads, _, _, scale = ctx.cfg[:4]
            T = qkv.shape[1]
            dqkv, dbias = K.window_attention_bwd(qkv, table, _c(dO), heads, scale)
            if ctx.bias_table.requires_grad:
                K.relpos_bias_scatter_grad(dbias, T, ctx.bias_index, _grad(ctx.bias_table), heads, T, accumulate=True,
                                           window=ctx.bias_window)
                dist.grads_ready(ctx.bias_table)
            return (dqkv,) + (None,) * 13
        qkv, P, Pd = ctx.saved_tensors
        heads, Cq, Cv, scale, windows, clip, drop_rate, seed, Tp = ctx.cfg
        B, T, ld = qkv.shape
        dq, dv = Cq // heads, Cv // heads
        dO = _c(dO)
        Pd = P if Pd is None else Pd
        dqkv = torch.empty_like(qkv)
        dP = torch.empty_like(P)
        qv, kv, vv = qkv[:, :, :Cq], qkv[:, :, Cq:2 * Cq], qkv[:, :, 2 * Cq:]
        dqv, dkv, dvv = dqkv[:, :, :Cq], dqkv[:, :, Cq:2 * Cq], dqkv[:, :, 2 * Cq:]
        sP = (heads * T * Tp, T * Tp)
        for b0, b1 in _attn_chunks(B, heads):
            nb = (b1 - b0) * heads
            z0, z1 = b0 * heads, b1 * heads
            # dP = dO V^T ; dV = P^T dO
            K.gemm(dO[b0:b1], vv[b0:b1], dP[z0:z1], T, T, dv, lda=Cv, ldb=ld, ldd=Tp, a_kcontig=1, b_kcontig=1, batch=nb,
                   batch_inner=heads, sa=(T * Cv, dv), sb=(T * ld, dv), sd=sP)
            K.gemm(Pd[z0:z1], dO[b0:b1], dvv[b0:b1], T, dv, T, lda=Tp, ldb=Cv, ldd=ld, a_kcontig=0, b_kcontig=0, batch=nb,
                   batch_inner=heads, sa=sP, sb=(T * Cv, dv), sd=(T * ld, dv))
        if clip is not None:
            dP = K.clip_bwd(ctx.pre_clip, dP, clip[0], clip[1])
        if drop_rate > 0:
            dP = K.dropout(dP, drop_rate, seed)
        K.softmax_rows_bwd(P, dP, B * heads * T, T, Tp)
        if ctx.bias_table is not None and ctx.bias_table.requires_grad:
            dbias = torch.empty(heads * T * Tp, dtype=torch.float32, device=qkv.device)
            K.colsum_wide(dP.reshape(B, heads * T * Tp), dbias)
            K.relpos_bias_scatter_grad(dbias, Tp, ctx.bias_index, _grad(ctx.bias_table), heads, T, accumulate=True,
                                       window=ctx.bias_window)
            dist.grads_ready(ctx.bias_table)
        for b0, b1 in _attn_chunks(B, heads):
            nb = (b1 - b0) * heads
            z0, z1 = b0 * heads, b1 * heads
            # dQ = scale * dS K ; dK = scale * dS^T Q
            K.gemm(dP[z0:z1], kv[b0:b1], dqv[b0:b1], T, dq, T, lda=Tp, ldb=ld, ldd=ld, a_kcontig=1, b_kcontig=0, alpha=scale,
                   batch=nb, batch_inner=heads, sa=sP, sb=(T * ld, dq), sd=(T * ld, dq))
            K.gemm(dP[z0:z1], qv[b0:b1], dkv[b0:b1], T, dq, T, lda=Tp, ldb=ld, ldd=ld, a_kcontig=0, b_kcontig=0, alpha=scale,
                   batch=nb, batch_inner=heads, sa=sP, sb=(T * ld, dq), sd=(T * ld, dq))
        return (dqkv,) + (None,) * 13


def attention_packed(qkv, heads, Cq, Cv, scale, *, bias_table=None, bias_index=None, mask=None, windows=1, clip=None,
                     dropout_rate=0.0, training=False, bias_window=0, return_probs=False):
    """qkv [B, T, 2*Cq + Cv] (columns [q | k | v], each split into `heads` contiguous head slices) -> [B, T, Cv].
    bias_table [entries, heads] fp32 parameter + bias_index int32 [T*T]; mask fp32 [windows, T, T] (sample b uses mask
    b % windows); clip = (lo, hi) on the probabilities."""
    _check_act_dtype(qkv)
    if nn.dry_run():
        out = _dry((qkv.shape[0], qkv.shape[1], Cv), qkv)
        return (out, _dry((qkv.shape[0], heads, qkv.shape[1], qkv.shape[1]), qkv)) if return_probs else out
    if mask is not None and (_MAX_GRID_Z // heads) % windows != 0 and qkv.shape[0] * heads > _MAX_GRID_Z:
        raise NotImplementedError("attention_packed: chunked launch needs chunk sizes that are multiples of the window count")
    rate = float(dropout_rate) if training else 0.0
    if return_probs:
        return _AttentionFn.apply(qkv, bias_table, int(heads), int(Cq), int(Cv), float(scale), bias_index, mask, int(windows), clip, rate,
                                  next_seed() if rate > 0 else 0, int(bias_window), True)
    if (bias_table is None and mask is None and clip is None and rate <= 0 and Cq == Cv and
            K.attention_fwd_supported(Cq // heads, qkv.dtype) and os.environ.get("ISEG_FLASHATTN", "1") != "0"):
        # online-softmax kernels, no T x T tensor: forward only for inference, the recomputing forward / backward pair for training
        if not (torch.is_grad_enabled() and qkv.requires_grad):
            return K.attention_fwd(_c(qkv), int(heads), float(scale))
        return _FlashAttentionFn.apply(qkv, int(heads), float(scale))
    return _AttentionFn.apply(qkv, bias_table, int(heads), int(Cq), int(Cv), float(scale), bias_index, mask, int(windows), clip,
                              rate, next_seed() if rate > 0 else 0, int(bias_window))


class _Dcnv2SampleFn(Function):
    """DCNv2's modulated deformable sampling (layers/dcn_v2.py:114-229): x [N,H,W,C], offset [N,H,W,27] -> [N,H,W,9 C]"""

    @staticmethod
    def forward(ctx, x, offset):
        xc, oc = _c(x), _c(offset)
        ctx.save_for_backward(xc, oc)
        N, H, W, C = xc.shape
        return K.dcnv2_sample_fwd(xc, oc).reshape(N, H, W, 9 * C)

    @staticmethod
    def backward(ctx, dcol):
        xc, oc = ctx.saved_tensors
        dx, doff = K.dcnv2_sample_bwd(xc, oc, _c(dcol))
        return (dx if xc.dtype == torch.float32 else K.cast(dx, xc.dtype)), doff


def dcnv2_sample(x, offset):
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry((*x.shape[:3], 9 * x.shape[3]), x)
    return _Dcnv2SampleFn.apply(x, offset)


class _QkvRopeFn(Function):
    """packed attention rows [B, T, 3C]: q += q_bias, v += v_bias, rotary embedding on q and k of the tokens >= prefix -- one pass
    (csrc/eva.hip; backbones/eva/attention.py:100-112,136-146).  Backward: the transposed rotation in place on the gradient, bias gradients as
    strided column sums of its q / v blocks."""

    @staticmethod
    def forward(ctx, qkv, q_bias, v_bias, emb, prefix, heads):
        B, T, C3 = qkv.shape
        C = C3 // 3
        ctx.q_bias, ctx.v_bias, ctx.emb, ctx.geom = q_bias, v_bias, emb, (T, int(prefix), C, C // heads)
        src = _c(qkv)
        out = torch.empty_like(src)
        K.qkv_rope(src, None if q_bias is None else q_bias.data, None if v_bias is None else v_bias.data, emb, T, int(prefix), C, C // heads, out=out)
        return out

    @staticmethod
    def backward(ctx, d):
        T, prefix, C, hd = ctx.geom
        d = _c(d)
        g = torch.empty_like(d)
        K.qkv_rope(d, None, None, ctx.emb, T, prefix, C, hd, inverse=True, out=g)
        d2 = g.reshape(-1, 3 * C)
        for b, off in ((ctx.q_bias, 0), (ctx.v_bias, 2 * C)):
            if b is not None and b.requires_grad:
                K.colsum(d2[:, off:off + C], d2.stride(0), 0, 1, d2.shape[0], C, _grad(b).reshape(-1), accumulate=True)
        dist.grads_ready(*[b for b in (ctx.q_bias, ctx.v_bias) if b is not None])
        return g, None, None, None, None, None


def qkv_rope(qkv, q_bias, v_bias, emb, prefix, heads):
    """EVA attention's bias + rotary step on packed [B, T, 3C] rows (emb fp32 [T - prefix, 2 head_dim] or None)"""
    if nn.dry_run():
        return qkv
    if q_bias is None and v_bias is None and emb is None:
        return qkv
    return _QkvRopeFn.apply(qkv, q_bias, v_bias, emb, prefix, heads)


class _GluFn(Function):
    """act(gate) * x (backbones/eva/swiglu.py:88-92, glumlp.py:96-103); `packed` [.., 2H] = [x | gate] (gate_last) or [gate | x]"""

    @staticmethod
    def forward(ctx, gate, x, act, packed_order):
        if packed_order:      # gate is the packed tensor, x unused
            H = gate.shape[-1] // 2
            p2 = _c(gate).reshape(-1, 2 * H)
            g2, x2 = (p2[:, H:], p2[:, :H]) if packed_order == 1 else (p2[:, :H], p2[:, H:])
            out_shape = (*gate.shape[:-1], H)
        else:
            H = gate.shape[-1]
            g2, x2 = _c(gate).reshape(-1, H), _c(x).reshape(-1, H)
            out_shape = gate.shape
        ctx.act, ctx.packed_order, ctx.in_shape = act, packed_order, gate.shape
        ctx.save_for_backward(g2, x2)
        return K.glu_fwd(g2, x2, act).reshape(out_shape)

    @staticmethod
    def backward(ctx, dout):
        g2, x2 = ctx.saved_tensors
        H = g2.shape[1]
        d2 = _c(dout).reshape(-1, H)
        if ctx.packed_order:
            dp = torch.empty((g2.shape[0], 2 * H), dtype=g2.dtype, device=g2.device)
            dg, dx = (dp[:, H:], dp[:, :H]) if ctx.packed_order == 1 else (dp[:, :H], dp[:, H:])
            K.glu_bwd(d2, g2, x2, dg, dx, ctx.act)
            return dp.reshape(ctx.in_shape), None, None, None
        dg, dx = torch.empty_like(g2), torch.empty_like(x2)
        K.glu_bwd(d2, g2, x2, dg, dx, ctx.act)
        return dg.reshape(ctx.in_shape), dx.reshape(ctx.in_shape), None, None


_GLU_ACTS = {"gelu": K.ACT_GELU, "swish": K.ACT_SWISH, "silu": K.ACT_SWISH, "sigmoid": K.ACT_SIGMOID}


def glu(gate, x, activation):
    """activation(gate) * x, both [..., H]"""
    _check_act_dtype(gate)
    if nn.dry_run():
        return _dry(gate.shape, gate)
    return _GluFn.apply(gate, x, _GLU_ACTS[activation], 0)


def glu_packed(packed, activation, gate_last=True):
    """x1, x2 = split(packed, 2, axis=-1); x1 * activation(x2) if gate_last else activation(x1) * x2   (glumlp.py:96-103)"""
    _check_act_dtype(packed)
    if nn.dry_run():
        return _dry((*packed.shape[:-1], packed.shape[-1] // 2), packed)
    return _GluFn.apply(packed, None, _GLU_ACTS[activation], 1 if gate_last else 2)


class _GatherRowsFn(Function):
    @staticmethod
    def forward(ctx, x, idx_fwd, idx_bwd, out_shape):
        C = x.shape[-1]
        ctx.in_shape, ctx.idx_bwd = x.shape, idx_bwd
        return K.gather_rows(_c(x).reshape(-1, C), idx_fwd, idx_fwd.numel()).reshape(out_shape)

    @staticmethod
    def backward(ctx, dy):
        C = dy.shape[-1]
        idx = ctx.idx_bwd
        return K.gather_rows(_c(dy).reshape(-1, C), idx, idx.numel()).reshape(ctx.in_shape), None, None, None


def permute_rows(x, idx_fwd, idx_bwd, out_shape):
    """rows of x (last axis = channels) re-ordered by a static index table; idx_bwd is the inverse table (-1 where a source
    row has no destination / a destination is padding).  Each source row may appear at most once in idx_fwd."""
    if nn.dry_run():
        return _dry(out_shape, x)
    return _GatherRowsFn.apply(x, idx_fwd, idx_bwd, tuple(out_shape))


class _LnGatherFn(Function):
    """LayerNorm + static row permutation with zero padding in one pass each way (Swin: norm1 + pad + roll + window partition).  `link`
    pairs the node with the _GatherResidualFn that closes the same residual branch: that node's backward runs first and leaves the skip
    connection's gradient in link["dres"], which this node's LayerNorm backward adds on its way out (dx_add) -- the branch's fork costs no pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, idx_fwd, idx_bwd, out_shape, link):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        y, mean, rstd = K.layernorm_gather_fwd(x2, idx_fwd, gamma.data, beta.data, eps)
        ctx.gamma, ctx.beta, ctx.idx_bwd, ctx.link, ctx.in_shape = gamma, beta, idx_bwd, link, x.shape
        ctx.save_for_backward(x2, mean, rstd)
        if link is not None:
            link.pop("dres", None)      # nothing of an earlier step (a backward pass that stopped between the two nodes) survives into this one
            link["armed"] = bool(ctx.needs_input_grad[0])
            link["x"] = (x.data_ptr(), tuple(x.shape))      # the skip connection handed over through the link must be THIS tensor
        return y.reshape(out_shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        C = x2.shape[1]
        dres = ctx.link.pop("dres", None) if ctx.link is not None else None
        if dres is not None:
            dres = _c(dres).reshape(-1, C)
        dx = K.layernorm_gather_bwd(_c(dy).reshape(-1, C), ctx.idx_bwd, x2, ctx.gamma.data, mean, rstd, _grad(ctx.gamma), _grad(ctx.beta), dx_add=dres)
        dist.grads_ready(ctx.gamma, ctx.beta)
        return (dx.reshape(ctx.in_shape) if ctx.needs_input_grad[0] else None,) + (None,) * 7


class _GatherResidualFn(Function):
    """residual + rowscale[sample] * rows(y)[idx_fwd] in one pass (Swin: window reverse + roll back + crop + drop path + skip connection)"""

    @staticmethod
    def forward(ctx, y, residual, idx_fwd, idx_bwd, rowscale, link):
        C = y.shape[-1]
        r2 = _c(residual).reshape(-1, C)
        rpg = r2.shape[0] // rowscale.shape[0] if rowscale is not None else 0
        ctx.idx_bwd, ctx.rowscale, ctx.rpg, ctx.link, ctx.y_shape = idx_bwd, rowscale, rpg, link, y.shape
        if link is not None and link.get("armed") and link.get("x") != (residual.data_ptr(), tuple(residual.shape)):
            raise ValueError("permute_rows_residual: `link` pairs this node with a layer_norm_permute_rows of ANOTHER tensor -- the skip connection's "
                             "gradient would be added to the wrong LayerNorm input")
        return K.gather_rows_fma(_c(y).reshape(-1, C), idx_fwd, rowscale, rpg, False, r2).reshape(residual.shape)

    @staticmethod
    def backward(ctx, dout):
        C = dout.shape[-1]
        d2 = _c(dout).reshape(-1, C)
        dy = K.gather_rows_fma(d2, ctx.idx_bwd, ctx.rowscale, ctx.rpg, True, None).reshape(ctx.y_shape) if ctx.needs_input_grad[0] else None
        dres = dout if ctx.needs_input_grad[1] else None
        if dres is not None and ctx.link is not None and ctx.link.get("armed") and dy is not None:
            ctx.link["dres"] = dres      # the paired _LnGatherFn adds it inside its LayerNorm backward
            dres = None
        return dy, dres, None, None, None, None


def residual_branch_link():
    """the pairing object of layer_norm_permute_rows / permute_rows_residual on one residual branch (see _LnGatherFn)"""
    return {}


def layer_norm_permute_rows(x, gamma, beta, eps, idx_fwd, idx_bwd, out_shape, link=None):
    """permute_rows(layer_norm(x), idx_fwd, idx_bwd, out_shape) as one tape node; padding rows (idx_fwd < 0) are zero"""
    _check_act_dtype(x)
    if nn.dry_run():
        return _dry(out_shape, x)
    return _LnGatherFn.apply(x, gamma, beta, float(eps), idx_fwd, idx_bwd, tuple(out_shape), link)


def permute_rows_residual(y, residual, idx_fwd, idx_bwd, drop_path_mask=None, link=None):
    """residual + drop_path_mask[sample] * permute_rows(y, idx_fwd, idx_bwd, residual.shape) as one tape node.  With `link` shared with the
    layer_norm_permute_rows that opened the branch FROM THE SAME TENSOR `residual`, the skip connection's gradient is delivered through that
    node (the caller must not consume `residual` anywhere else between the two)."""
    if nn.dry_run():
        return _dry(residual.shape, residual)
    return _GatherResidualFn.apply(y, residual, idx_fwd, idx_bwd, drop_path_mask, link)


# ---------------------------------------------------------------------------------------------------------
# token-axis helpers of backbones/vit.py:277-323: class token, position embedding, token slicing
# ---------------------------------------------------------------------------------------------------------
class _PrependTokenFn(Function):
    @staticmethod
    def forward(ctx, x, token):
        B, T, C = x.shape
        xc = _c(x)
        y = torch.empty((B, T + 1, C), dtype=x.dtype, device=x.device)
        tok = K.cast(token.data.reshape(1, C), x.dtype)
        K.copy2d(tok, 0, y, (T + 1) * C, B, C)                          # row stride 0: the same token for every sample
        K.copy2d(xc, T * C, y.reshape(B, -1)[:, C:], (T + 1) * C, B, T * C)
        ctx.token = token
        return y

    @staticmethod
    def backward(ctx, dy):
        B, T1, C = dy.shape
        dyc = _c(dy)
        if ctx.token.requires_grad:
            K.colsum(dyc, T1 * C, 0, 1, B, C, _grad(ctx.token).reshape(-1), accumulate=True)
            dist.grads_ready(ctx.token)
        dx = torch.empty((B, T1 - 1, C), dtype=dy.dtype, device=dy.device)
        K.copy2d(dyc.reshape(B, -1)[:, C:], T1 * C, dx, (T1 - 1) * C, B, (T1 - 1) * C)
        return dx, None


def prepend_token(x, token):
    """tf.concat([broadcast(class_token), x], axis=1); token is an fp32 parameter [1,1,C]"""
    if nn.dry_run():
        return _dry((x.shape[0], x.shape[1] + 1, x.shape[2]), x)
    return _PrependTokenFn.apply(x, token)


class _DropTokensFn(Function):
    @staticmethod
    def forward(ctx, x, n_extra):
        B, T, C = x.shape
        ctx.n_extra, ctx.shape = n_extra, x.shape
        y = torch.empty((B, T - n_extra, C), dtype=x.dtype, device=x.device)
        K.copy2d(_c(x).reshape(B, -1)[:, n_extra * C:], T * C, y, (T - n_extra) * C, B, (T - n_extra) * C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, T, C = ctx.shape
        e = ctx.n_extra
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        zero = torch.zeros((1, e * C), dtype=dy.dtype, device=dy.device)
        K.copy2d(zero, 0, dx, T * C, B, e * C)
        K.copy2d(_c(dy), (T - e) * C, dx.reshape(B, -1)[:, e * C:], T * C, B, (T - e) * C)
        return dx, None


def drop_tokens(x, n_extra):
    """x[:, n_extra:]"""
    if n_extra == 0:
        return x
    if nn.dry_run():
        return _dry((x.shape[0], x.shape[1] - n_extra, x.shape[2]), x)
    return _DropTokensFn.apply(x, int(n_extra))


class _AddBatchBroadcastFn(Function):
    @staticmethod
    def forward(ctx, x, v):
        B = x.shape[0]
        n = x.numel() // B
        y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        K.copy2d(_c(x), n, y, n, B, n)
        K.broadcast_rows(_c(v).reshape(1, n), y, n, 0, 1, B, n, accumulate=True)
        ctx.v_dtype, ctx.v_shape = v.dtype, v.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        B = dy.shape[0]
        n = dy.numel() // B
        dv = torch.empty(n, dtype=torch.float32, device=dy.device)
        K.colsum(_c(dy), n, 0, 1, B, n, dv)
        dv = dv if ctx.v_dtype == torch.float32 else K.cast(dv, ctx.v_dtype)
        return dy, dv.reshape(ctx.v_shape)


def add_batch_broadcast(x, v):
    """x [B, ...] + v [1, ...] (tf.add with broadcasting over the batch axis: position embedding)"""
    if nn.dry_run():
        return _dry(x.shape, x)
    return _AddBatchBroadcastFn.apply(x, v)


class _PosEmbedResizeFn(Function):
    """backbones/vit.py:19-63 resize_pos_embed: the grid part [1, g*g, C] of the fp32 position embedding is resampled to
    (Ho, Wo) by tf.image.resize(bicubic) -- a separable linear map, expressed as two fp32 GEMMs y = Wy @ x @ Wx^T per channel
    with constant weight matrices built on the host (utils/bicubic.py) -- the extra (class) tokens are passed through, and the
    result is cast to the compute dtype.  The gradient is written straight into the parameter's gradient buffer."""

    @staticmethod
    def forward(ctx, pos, Wy, Wx, n_extra, out_dtype):
        _, L, C = pos.shape
        Hi, Wi = Wy.shape[1], Wx.shape[1]
        Ho, Wo = Wy.shape[0], Wx.shape[0]
        assert L == n_extra + Hi * Wi
        src = pos.data.reshape(L, C)
        out = torch.empty((1, n_extra + Ho * Wo, C), dtype=torch.float32, device=pos.device)
        o2 = out.reshape(-1, C)
        if n_extra:
            K.copy2d(src, C, o2, C, n_extra, C)
        grid = src[n_extra:]
        t = torch.empty((Ho, Wi * C), dtype=torch.float32, device=pos.device)
        K.gemm(Wy, grid, t, Ho, Wi * C, Hi, lda=Hi, ldb=Wi * C, ldd=Wi * C, a_kcontig=1, b_kcontig=0)
        K.gemm(Wx, t, o2[n_extra:], Wo, C, Wi, lda=Wi, ldb=C, ldd=C, a_kcontig=1, b_kcontig=0, batch=Ho, batch_inner=1,
               sa=(0, 0), sb=(Wi * C, 0), sd=(Wo * C, 0))
        ctx.pos, ctx.n_extra = pos, n_extra
        ctx.save_for_backward(Wy, Wx)
        return out if out_dtype == torch.float32 else K.cast(out, out_dtype)

    @staticmethod
    def backward(ctx, dy):
        Wy, Wx = ctx.saved_tensors
        pos, e = ctx.pos, ctx.n_extra
        if not pos.requires_grad:
            return None, None, None, None, None
        C = pos.shape[-1]
        Hi, Wi = Wy.shape[1], Wx.shape[1]
        Ho, Wo = Wy.shape[0], Wx.shape[0]
        d = _c(dy) if dy.dtype == torch.float32 else K.cast(_c(dy), torch.float32)
        d2 = d.reshape(-1, C)
        g = _grad(pos).reshape(-1, C)
        if e:
            K.add2d(d2, C, g, C, e, C)
        dt_ = torch.empty((Ho, Wi * C), dtype=torch.float32, device=dy.device)
        K.gemm(Wx, d2[e:], dt_, Wi, C, Wo, lda=Wi, ldb=C, ldd=C, a_kcontig=0, b_kcontig=0, batch=Ho, batch_inner=1, sa=(0, 0),
               sb=(Wo * C, 0), sd=(Wi * C, 0))
        K.gemm(Wy, dt_, g[e:], Hi, Wi * C, Ho, lda=Hi, ldb=Wi * C, ldd=Wi * C, a_kcontig=0, b_kcontig=0, accumulate=True)
        dist.grads_ready(pos)
        return None, None, None, None, None


_BICUBIC_MATRICES = {}


def _bicubic_weights(out_size, in_size, device):
    key = (int(out_size), int(in_size), str(device))
    m = _BICUBIC_MATRICES.get(key)
    if m is None:
        from .utils.bicubic import bicubic_matrix

        m = _BICUBIC_MATRICES[key] = torch.from_numpy(bicubic_matrix(int(out_size), int(in_size))).to(device)
    return m


class _ResizeBicubicFn(Function):
    """tf.image.resize(images, size, method="bicubic") (utils/common.py:107-134 of the reference: half-pixel centres, Keys a = -0.5, no
    antialias, fp32 result cast back to the input dtype): the separable map y[n] = Wy @ x[n] @ Wx^T as two strided-batch fp32 GEMMs with the
    host-built tap matrices of utils/bicubic.py (the kernel of the ViT position-embedding resize).  The gradient is the transposed map."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        N, Hi, Wi, C = x.shape
        xf = _c(x) if x.dtype == torch.float32 else K.cast(_c(x), torch.float32)
        Wy, Wx = _bicubic_weights(Ho, Hi, x.device), _bicubic_weights(Wo, Wi, x.device)
        t = torch.empty((N, Ho, Wi * C), dtype=torch.float32, device=x.device)
        K.gemm(Wy, xf.reshape(N, Hi, Wi * C), t, Ho, Wi * C, Hi, lda=Hi, ldb=Wi * C, ldd=Wi * C, a_kcontig=1, b_kcontig=0, batch=N, batch_inner=1,
               sa=(0, 0), sb=(Hi * Wi * C, 0), sd=(Ho * Wi * C, 0))
        y = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
        K.gemm(Wx, t, y, Wo, C, Wi, lda=Wi, ldb=C, ldd=C, a_kcontig=1, b_kcontig=0, batch=N * Ho, batch_inner=1, sa=(0, 0), sb=(Wi * C, 0),
               sd=(Wo * C, 0))
        ctx.shape, ctx.dtype = (N, Hi, Wi, C), x.dtype
        ctx.save_for_backward(Wy, Wx)
        return y if x.dtype == torch.float32 else K.cast(y, x.dtype)

    @staticmethod
    def backward(ctx, dy):
        Wy, Wx = ctx.saved_tensors
        N, Hi, Wi, C = ctx.shape
        Ho, Wo = Wy.shape[0], Wx.shape[0]
        d = _c(dy) if dy.dtype == torch.float32 else K.cast(_c(dy), torch.float32)
        dt_ = torch.empty((N, Ho, Wi * C), dtype=torch.float32, device=dy.device)
        K.gemm(Wx, d, dt_, Wi, C, Wo, lda=Wi, ldb=C, ldd=C, a_kcontig=0, b_kcontig=0, batch=N * Ho, batch_inner=1, sa=(0, 0), sb=(Wo * C, 0),
               sd=(Wi * C, 0))
        dx = torch.empty((N, Hi, Wi * C), dtype=torch.float32, device=dy.device)
        K.gemm(Wy, dt_, dx, Hi, Wi * C, Ho, lda=Hi, ldb=Wi * C, ldd=Wi * C, a_kcontig=0, b_kcontig=0, batch=N, batch_inner=1, sa=(0, 0),
               sb=(Ho * Wi * C, 0), sd=(Hi * Wi * C, 0))
        dx = dx.reshape(N, Hi, Wi, C)
        return (dx if ctx.dtype == torch.float32 else K.cast(dx, ctx.dtype)), None, None


def resize_bicubic(x, size):
    Ho, Wo = int(size[0]), int(size[1])
    if nn.dry_run():
        return _dry((x.shape[0], Ho, Wo, x.shape[3]), x)
    return _ResizeBicubicFn.apply(x, Ho, Wo)


def resize_pos_embed(pos, Wy, Wx, n_extra, out_dtype):
    if nn.dry_run():
        return _dry((1, n_extra + Wy.shape[0] * Wx.shape[0], pos.shape[-1]), pos, out_dtype)
    return _PosEmbedResizeFn.apply(pos, Wy, Wx, int(n_extra), out_dtype)


# ---------------------------------------------------------------------------------------------------------
# DCNv3 core (layers/dcn_v3/op.py:16-109), softmax over the last axis in groups, per-channel scale
# ---------------------------------------------------------------------------------------------------------
class _Dcnv3Fn(Function):
    @staticmethod
    def forward(ctx, x, offset, mask, cfg):
        G, Cg, kh, kw, stride, dil, pad, s = cfg
        xc, oc, mc = _c(x), _c(offset), _c(mask)
        ctx.cfg = cfg
        ctx.save_for_backward(xc, oc, mc)
        return K.dcnv3_fwd(xc, oc, mc, G, Cg, kh, kw, stride, dil, pad, s)

    @staticmethod
    def backward(ctx, dy):
        xc, oc, mc = ctx.saved_tensors
        G, Cg, kh, kw, stride, dil, pad, s = ctx.cfg
        dx, doff, dmask = K.dcnv3_bwd(xc, oc, mc, _c(dy), G, Cg, kh, kw, stride, dil, pad, s)
        if dx.dtype != xc.dtype:
            dx = K.cast(dx, xc.dtype)
        return dx, doff, dmask, None


_DCN_JOINT = os.environ.get("ISEG_DCN_JOINT", "1") != "0"      # experiment knob: 0 = offset and mask as two Dense layers + softmax_groups


class _DcnJointFn(Function):
    """offset | mask projection, mask softmax and the DCNv3 sampling core of one layer (layers/dcn_v3/dcn_v3.py:116-131) with the two Dense layers
    run as ONE product of width ld = 3 G P rounded up to 8 into a [pixels, ld] matrix whose column ranges the sampling kernels read in place
    (csrc/dcnv3.hip iseg_dcnv3_fwd_ld): widths like 126 / 63 (G = 7) or 252 (G = 28) keep two products off the LDS-DMA GEMM each way, their sum
    padded to 192 / 768 does not.  Backward: sampling gradients in the same layout, softmax backward in place, ONE data-gradient product, ONE
    weight-gradient product whose [C, ld] result (and ones-row bias gradient) is added column-range-wise into the two layers' gradients."""

    @staticmethod
    def forward(ctx, x_proj, x1, W_off, b_off, W_mask, b_mask, cfg):
        G, Cg, kh, kw, stride, dil, pad, s = cfg
        P = kh * kw
        Cin = x1.shape[-1]
        x2 = _c(x1).reshape(-1, Cin)
        rowcat, tr, bias = nn.joint_kernels(W_off, W_mask, b_off, b_mask)
        om = K.dense_fwd_t(x2, tr, bias)
        K.dcn_mask_softmax_fwd(om, G, P)
        xp = _c(x_proj)
        y = K.dcnv3_fwd_joint(xp, om, G, Cg, kh, kw, stride, dil, pad, s)
        ctx.cfg, ctx.params = cfg, (W_off, b_off, W_mask, b_mask)
        ctx.save_for_backward(xp, x2, om)
        ctx.x1_shape = x1.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, x2, om = ctx.saved_tensors
        G, Cg, kh, kw, stride, dil, pad, s = ctx.cfg
        P = kh * kw
        W_off, b_off, W_mask, b_mask = ctx.params
        dxp, dom = K.dcnv3_bwd_joint(xp, om, _c(dy), G, Cg, kh, kw, stride, dil, pad, s)
        if dxp.dtype != xp.dtype:
            dxp = K.cast(dxp, xp.dtype)
        K.dcn_mask_softmax_bwd(om, dom, G, P)
        rowcat, tr, bias = nn.joint_kernels(W_off, W_mask, b_off, b_mask)
        Cin, ld = rowcat.shape
        na, nb = 2 * G * P, G * P
        want_w = W_off.requires_grad or W_mask.requires_grad
        want_b = b_off.requires_grad or b_mask.requires_grad
        if want_w:
            dW = torch.empty((Cin, ld), dtype=torch.float32, device=dom.device)
            db = torch.empty(ld, dtype=torch.float32, device=dom.device) if want_b else None
            K.dense_wgrad(x2, dom, dW, accumulate=False, bias_grad=db)
            K.split_cols_accumulate(dW, _grad(W_off).reshape(Cin, na) if W_off.requires_grad else None, na,
                                    _grad(W_mask).reshape(Cin, nb) if W_mask.requires_grad else None, nb)
        elif want_b:
            db = torch.empty(ld, dtype=torch.float32, device=dom.device)
            K.colsum(dom, ld, 0, 1, dom.shape[0], ld, db, accumulate=False)
        if want_b:
            K.split_cols_accumulate(db, _grad(b_off).reshape(-1) if b_off.requires_grad else None, na,
                                    _grad(b_mask).reshape(-1) if b_mask.requires_grad else None, nb)
        dx1 = None
        if ctx.needs_input_grad[1]:
            dx1 = K.dense_dgrad(dom, rowcat).reshape(ctx.x1_shape)
        dist.grads_ready(W_off, b_off, W_mask, b_mask)
        return dxp, dx1, None, None, None, None, None


def dcnv3_joint_ok(x1, offset_layer, mask_layer, kernel_size, stride=1, out_hw=None):
    """stride / out_hw: the joint form projects x1 pixel by pixel, so the offset / mask rows are x1's pixels -- they are the sampling kernel's OUTPUT
    pixels only at stride 1 with an unchanged map size (round-5 advisor: a strided layer must fall back to the layer-by-layer route, not raise)"""
    kh, kw = kernel_size
    if int(stride) != 1 or (out_hw is not None and tuple(out_hw) != tuple(x1.shape[1:3])):
        return False
    return (_DCN_JOINT and not nn.dry_run() and x1.dtype == torch.bfloat16 and kh * kw <= 9 and getattr(offset_layer, "kernel", None) is not None
            and getattr(mask_layer, "kernel", None) is not None and offset_layer.bias is not None and mask_layer.bias is not None
            and x1.shape[-1] % 8 == 0 and x1.shape[-1] >= 64)


def dcnv3_joint(x_proj, x1, offset_layer, mask_layer, groups, group_channels, kernel_size=(3, 3), stride=1, dilation=1, pad=1, offset_scale=1.0):
    """F.dcnv3_core(x_proj, offset_layer(x1), softmax_groups(mask_layer(x1)), ...) as one node (see _DcnJointFn), or None when this form does not
    apply (fp32 storage, more than nine sampling points, a projection without bias, ISEG_DCN_JOINT=0): the caller then takes the layer-by-layer
    route"""
    kh, kw = kernel_size
    if not dcnv3_joint_ok(x1, offset_layer, mask_layer, kernel_size, stride, K.dcnv3_out_hw(x_proj.shape[1], x_proj.shape[2], kh, kw, stride, dilation, pad)):
        return None
    cfg = (int(groups), int(group_channels), int(kh), int(kw), int(stride), int(dilation), int(pad), float(offset_scale))
    return _DcnJointFn.apply(x_proj, x1, offset_layer.kernel, offset_layer.bias, mask_layer.kernel, mask_layer.bias, cfg)


class _DcnCenterBlendFn(Function):
    @staticmethod
    def forward(ctx, x, x_proj, scale, G, Cg):
        xc, pc, sc = _c(x), _c(x_proj), _c(scale)
        ctx.save_for_backward(xc, pc, sc)
        ctx.geom = (G, Cg)
        return K.dcn_center_blend_fwd(xc, pc, sc, G, Cg)

    @staticmethod
    def backward(ctx, dout):
        xc, pc, sc = ctx.saved_tensors
        dx, dxp, ds = K.dcn_center_blend_bwd(_c(dout), xc, pc, sc, *ctx.geom)
        return dx, dxp, ds, None, None


def dcn_center_blend(x, x_proj, scale, groups, group_channels):
    """centre-feature scale of the DCNv3 layer (layers/dcn_v3/dcn_v3.py:138-146): x (1 - s) + x_proj s, s [N, H, W, groups] broadcast over
    the channels of its group"""
    _check_act_dtype(x)
    if group_channels not in (8, 16):
        raise NotImplementedError("dcn_center_blend: group widths 8 and 16 (the reference's models use 16)")
    if nn.dry_run():
        return _dry(x.shape, x)
    return _DcnCenterBlendFn.apply(x, x_proj, scale, int(groups), int(group_channels))


def dcnv3_core(x, offset, mask, groups, group_channels, kernel_size=(3, 3), stride=1, dilation=1, pad=1, offset_scale=1.0):
    kh, kw = kernel_size
    if nn.dry_run():
        Ho, Wo = K.dcnv3_out_hw(x.shape[1], x.shape[2], kh, kw, stride, dilation, pad)
        return _dry((x.shape[0], Ho, Wo, x.shape[3]), x)
    return _Dcnv3Fn.apply(x, offset, mask, (int(groups), int(group_channels), int(kh), int(kw), int(stride), int(dilation), int(pad),
                                             float(offset_scale)))


class _GroupSoftmaxFn(Function):
    @staticmethod
    def forward(ctx, x, P):
        xc = _c(x)
        rows = xc.numel() // P
        y = torch.empty_like(xc)
        K.softmax_rows_fwd(xc, rows, 1, P, P, out=y)
        ctx.P = P
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        P = ctx.P
        d = torch.empty_like(y)
        K.softmax_rows_bwd(y, _c(dy), y.numel() // P, P, P, out=d)
        return d, None


def softmax_groups(x, P):
    """softmax over consecutive runs of P entries of the last axis (tf.reshape [..., G, P] -> tf.nn.softmax -> reshape back)"""
    if nn.dry_run():
        return _dry(x.shape, x)
    return _GroupSoftmaxFn.apply(x, int(P))


class _ScaleChannelsFn(Function):
    @staticmethod
    def forward(ctx, x, gamma):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        ctx.gamma = gamma
        ctx.save_for_backward(x2)
        return K.scale_cols(x2, gamma.data).reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        g = ctx.gamma
        C = x2.shape[1]
        dy2 = _c(dy).reshape(-1, C)
        if g.requires_grad:
            K.mul_colsum(dy2, x2, _grad(g), accumulate=True)
            dist.grads_ready(g)
        return K.scale_cols(dy2, g.data).reshape(dy.shape), None


def scale_channels(x, gamma):
    """x * gamma[c] with an fp32 parameter gamma (layer scale)"""
    if nn.dry_run():
        return _dry(x.shape, x)
    return _ScaleChannelsFn.apply(x, gamma)


_FLIP_CACHE = {}


def flip_left_right(x):
    """tf.image.flip_left_right on [N,H,W,C]: a row permutation (its own inverse)"""
    N, H, W, C = x.shape
    if nn.dry_run():
        return _dry(x.shape, x)
    key = (N, H, W, str(x.device))
    if key not in _FLIP_CACHE:
        base = torch.arange(N * H, dtype=torch.int32).reshape(-1, 1) * W
        _FLIP_CACHE[key] = (base + torch.arange(W - 1, -1, -1, dtype=torch.int32).reshape(1, -1)).reshape(-1).to(x.device)
    idx = _FLIP_CACHE[key]
    return permute_rows(x, idx, idx, tuple(x.shape))
