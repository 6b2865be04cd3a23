"""replace_nan_or_inf, GroupNormalization, RMSNormalization, SAME pooling (csrc/misc.hip) and the FPN / SimpleDecoder layers
built from them, vs the CPU oracle (forward and every gradient)."""
import pytest
import torch

from oracle import models as OM
from oracle import tf_ops as O
from tests.test_kernels_gpu import DTYPES, close, q, rnd
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu


def K():
    from iseg_amd import kernels

    return kernels


@pytest.mark.parametrize("dtype", DTYPES)
def test_replace_nan_or_inf(cuda, dtype):
    k = K()
    x = rnd((3, 5, 7, 16), 1).float()
    x[0, 0, 0, 0] = float("nan")
    x[1, 2, 3, 4] = float("inf")
    x[2, 4, 6, 15] = float("-inf")
    xs = x.to(dtype)
    y = k.replace_nan_or_inf(xs.cuda(), 0.0)
    want = O.replace_nan_or_inf(xs.double(), 0.0)
    assert torch.equal(y.cpu().double(), want)
    # no special values: identity, bit for bit
    clean = rnd((1000,), 2).to(dtype)
    assert torch.equal(k.replace_nan_or_inf(clean.cuda()).cpu(), clean)
    dy = rnd(x.shape, 3).to(dtype)
    dx = k.replace_nan_or_inf_bwd(xs.cuda(), dy.cuda()).cpu()
    mask = torch.isfinite(xs)
    assert torch.equal(dx, torch.where(mask, dy, torch.zeros_like(dy)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,groups", [((2, 8, 8, 64), 16), ((3, 5, 7, 96), 2), ((1, 16, 16, 768), 16), ((2, 3, 3, 32), 32),
                                          ((2, 4, 4, 24), 1)])
def test_groupnorm(cuda, dtype, shape, groups):
    k = K()
    N, H, W, C = shape
    x, xr = q(rnd(shape, 1) * 2 + 0.5, dtype)
    gamma, beta = (rnd((C,), 2) * 0.2 + 1).float(), (rnd((C,), 3) * 0.1).float()
    dy, dyr = q(rnd(shape, 4), dtype)
    y, mean, rstd = k.groupnorm_fwd(x.reshape(N, H * W, C), gamma.cuda(), beta.cuda(), groups, 1e-3)
    xr.requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = O.group_norm(xr, gr, br, groups, 1e-3)
    close(y.reshape(shape), yr, dtype, "groupnorm fwd", f32_tol=2e-5, bf16_tol=1.2e-2)
    yr.backward(dyr)
    dg = torch.zeros(C, device="cuda")
    db = torch.zeros(C, device="cuda")
    dx = k.groupnorm_bwd(dy.reshape(N, H * W, C), x.reshape(N, H * W, C), gamma.cuda(), mean, rstd, groups, dg, db, accumulate=False)
    close(dx.reshape(shape), xr.grad, dtype, "groupnorm dx", f32_tol=5e-5, bf16_tol=2e-2)
    close(dg, gr.grad, dtype, "groupnorm dgamma", f32_tol=5e-5, bf16_tol=2e-2)
    close(db, br.grad, dtype, "groupnorm dbeta", f32_tol=5e-5, bf16_tol=2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(37, 96), (1000, 768), (5, 100), (64, 2048)])
def test_rmsnorm(cuda, dtype, rows, C):
    k = K()
    x, xr = q(rnd((rows, C), 1), dtype)
    scale = (rnd((C,), 2) * 0.3).float()
    dy, dyr = q(rnd((rows, C), 3), dtype)
    y, rstd = k.rmsnorm_fwd(x, scale.cuda(), 1e-6)
    xr.requires_grad_(True)
    sr = scale.double().requires_grad_(True)
    yr = O.rms_norm(xr, sr, 1e-6)
    close(y, yr, dtype, "rmsnorm fwd")
    yr.backward(dyr)
    ds = torch.zeros(C, device="cuda")
    dx = k.rmsnorm_bwd(dy, x, scale.cuda(), rstd, ds, accumulate=False)
    close(dx, xr.grad, dtype, "rmsnorm dx", f32_tol=5e-5, bf16_tol=2e-2)
    close(ds, sr.grad, dtype, "rmsnorm dscale", f32_tol=5e-5, bf16_tol=2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["max", "avg"])
@pytest.mark.parametrize("shape,k_,s", [((2, 16, 16, 8), 3, 2), ((1, 9, 7, 16), 3, 2), ((2, 8, 8, 32), 2, 2), ((1, 5, 5, 8), 2, 2),
                                        ((1, 7, 10, 8), 3, 1)])
def test_pool_same(cuda, dtype, mode, shape, k_, s):
    from iseg_amd import functional as F
    from iseg_amd import nn

    nn.set_compute_dtype(dtype)
    try:
        x, xr = q(rnd(shape, 1), dtype)
        xg = x.requires_grad_(True)
        y = (F.max_pool2d if mode == "max" else F.avg_pool2d)(xg, k_, s, "same")
        xr.requires_grad_(True)
        yr = (O.max_pool_same if mode == "max" else O.avg_pool_same)(xr, k_, s)
        assert tuple(y.shape) == tuple(yr.shape)
        close(y, yr, dtype, f"{mode} pool fwd", f32_tol=1e-6, bf16_tol=8e-3)
        dy, dyr = q(rnd(tuple(yr.shape), 2), dtype)
        y.backward(dy)
        yr.backward(dyr)
        close(xg.grad, xr.grad, dtype, f"{mode} pool bwd", f32_tol=1e-6, bf16_tol=1.2e-2)
    finally:
        nn.set_compute_dtype(torch.float32)


def test_max_pool_gradient_goes_to_first_maximum(cuda):
    k = K()
    x = torch.zeros(1, 4, 4, 8)          # every window is a tie: the first cell in row-major window order wins
    dy = torch.ones(1, 2, 2, 8)
    dx = k.pool2d_bwd(x.cuda(), dy.cuda(), 3, 3, 2, 2, 0, 0, k.POOL_MAX).cpu()
    want = torch.zeros(1, 4, 4, 8)
    for oh in range(2):
        for ow in range(2):
            want[0, oh * 2, ow * 2] += 1
    assert torch.equal(dx, want)


def _setup(layer, build_inputs, dtype):
    from iseg_amd import nn
    from iseg_amd.param_store import ParamStore

    with nn.dry_run_scope():
        layer(build_inputs)
    store = ParamStore(list(layer.parameters()))
    layer._iseg_store = store
    randomize_parameters(layer, 5)
    return store


def _param_err(got, ref, gmax):
    d = got.detach().cpu().double() - ref
    if got.dtype == torch.float32 and nn_is_bf16():
        return d.norm().item() / max(ref.norm().item(), 1e-3 * gmax * ref.numel() ** 0.5)
    return d.abs().max().item() / max(ref.abs().max().item(), 1e-3 * gmax)


def nn_is_bf16():
    from iseg_amd import nn

    return nn.compute_dtype() == torch.bfloat16


def _rel(a, b):
    """fp32: max-norm relative error.  bf16: relative L2 error -- a ReLU whose pre-activation rounds across zero in bf16
    flips single gradient entries by O(1), which a max-norm would report as failure of an otherwise exact operator."""
    a = a.detach().cpu()
    d = a.double() - b
    if a.dtype == torch.bfloat16:
        return d.norm().item() / max(b.norm().item(), 1e-8)
    return d.abs().max().item() / max(b.abs().max().item(), 1e-8)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("training", [False, True])
def test_fpn_layer(cuda, dtype, training):
    from iseg_amd import nn
    from iseg_amd.layers.fpn import FeaturePyramidNetwork

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        shapes = [(2, 16, 16, 24), (2, 8, 8, 48), (2, 4, 4, 96), (2, 2, 2, 64)]
        fpn = FeaturePyramidNetwork(skip_conv_filters=64, name="fpn")
        _setup(fpn, [torch.empty(s, dtype=dtype, device="cuda") for s in shapes], dtype)
        feats = [rnd(s, 10 + i).to(dtype) for i, s in enumerate(shapes)]
        fg = [f.cuda().requires_grad_(True) for f in feats]
        from iseg_amd import functional as F

        seen = []
        orig = F._BnReluUpsampleAddFn.apply
        F._BnReluUpsampleAddFn.apply = staticmethod(lambda *a: (seen.append(tuple(a[0].shape)), orig(*a))[1])
        try:
            outs = fpn(fg, training=training)
        finally:
            del F._BnReluUpsampleAddFn.apply
        # training mode: BatchNorm + ReLU + up-sampling + sum of every level is ONE node (csrc/resize.hip bn_relu_upsample_add_kernel)
        assert seen == ([(2, 4, 4, 64), (2, 8, 8, 64), (2, 16, 16, 64)] if training else []), seen
        w = {k: v.requires_grad_(True) for k, v in OM.export_weights(fpn).items()}
        fr = [f.double().requires_grad_(True) for f in feats]
        outs_r = OM.fpn(w, "fpn", fr, training)
        tol = 2e-4 if dtype == torch.float32 else 4e-2
        for a, b in zip(outs, outs_r):
            assert _rel(a, b.detach()) < tol
        dys = [rnd(tuple(o.shape), 20 + i).to(dtype) for i, o in enumerate(outs_r)]
        torch.autograd.backward(list(outs), [d.cuda() for d in dys])
        torch.autograd.backward(outs_r, [d.double() for d in dys])
        errs = {f"dfeat{i}": _rel(a.grad, b.grad) for i, (a, b) in enumerate(zip(fg, fr))}
        gmax = max(w[p.iseg_name].grad.abs().max().item() for p in fpn.parameters())
        for p in fpn.parameters():
            ref = w[p.iseg_name].grad
            errs[p.iseg_name] = _param_err(p.grad, ref, gmax)
        bad = {k: v for k, v in errs.items() if v > (5e-4 if dtype == torch.float32 else 0.15)}
        assert not bad, bad
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_simple_decoder_layer(cuda, dtype):
    from iseg_amd import nn
    from iseg_amd.layers.simpledecoder import SimpleDecoder

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        low_s, high_s = (2, 12, 10, 32), (2, 3, 3, 64)
        dec = SimpleDecoder(low_level_filters=16, mlp_filters=32, name="dec")
        _setup(dec, (torch.empty(low_s, dtype=dtype, device="cuda"), torch.empty(high_s, dtype=dtype, device="cuda")), dtype)
        low, high = rnd(low_s, 1).to(dtype), rnd(high_s, 2).to(dtype)
        lg, hg = low.cuda().requires_grad_(True), high.cuda().requires_grad_(True)
        y = dec((lg, hg), training=True)
        w = {k: v.requires_grad_(True) for k, v in OM.export_weights(dec).items()}
        lr_, hr_ = low.double().requires_grad_(True), high.double().requires_grad_(True)
        yr = OM.simple_decoder(w, "dec", lr_, hr_, True)
        tol = 2e-4 if dtype == torch.float32 else 4e-2
        assert _rel(y, yr.detach()) < tol
        dy = rnd(tuple(yr.shape), 3).to(dtype)
        y.backward(dy.cuda())
        yr.backward(dy.double())
        errs = {"dlow": _rel(lg.grad, lr_.grad), "dhigh": _rel(hg.grad, hr_.grad)}
        gmax = max(w[p.iseg_name].grad.abs().max().item() for p in dec.parameters())
        for p in dec.parameters():
            ref = w[p.iseg_name].grad
            errs[p.iseg_name] = _param_err(p.grad, ref, gmax)
        bad = {k: v for k, v in errs.items() if v > (5e-4 if dtype == torch.float32 else 0.15)}
        assert not bad, bad
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_groupnorm_layer_through_factory(cuda, dtype):
    from iseg_amd import nn
    from iseg_amd.layers.model_builder import ConvNormAct
    from iseg_amd.layers.normalizations import GROUP_NROM, normalization
    import functools

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        cna = ConvNormAct(32, (3, 3), norm_func=functools.partial(normalization, method=GROUP_NROM, groups=4), name="cna")
        _setup(cna, torch.empty((2, 6, 6, 16), dtype=dtype, device="cuda"), dtype)
        x = rnd((2, 6, 6, 16), 1).to(dtype)
        xg = x.cuda().requires_grad_(True)
        y = cna(xg, training=True)
        w = {k: v.requires_grad_(True) for k, v in OM.export_weights(cna).items()}
        xr = x.double().requires_grad_(True)
        yr = torch.relu(O.group_norm(O.conv2d(xr, w["cna/conv/kernel"], None, 1, 1, "same"), w["cna/bn/gamma"], w["cna/bn/beta"], 4, 1e-3))
        assert _rel(y, yr.detach()) < (2e-4 if dtype == torch.float32 else 4e-2)
        dy = rnd(tuple(yr.shape), 2).to(dtype)
        y.backward(dy.cuda())
        yr.backward(dy.double())
        tol = 5e-4 if dtype == torch.float32 else 0.15
        assert _rel(xg.grad, xr.grad) < tol
        gmax = max(w[p.iseg_name].grad.abs().max().item() for p in cna.parameters())
        for p in cna.parameters():
            assert _param_err(p.grad, w[p.iseg_name].grad, gmax) < tol, p.iseg_name
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,size", [((2, 5, 7, 16), (13, 20)), ((1, 8, 8, 48), (32, 32)), ((2, 9, 4, 5), (3, 2)), ((1, 1, 6, 8), (4, 1)),
                                        ((2, 16, 16, 24), (16, 31))])
def test_resize_bilinear_align_corners_matches_oracle(cuda, dtype, shape, size):
    """tf.compat.v1.image.resize(..., method="bilinear", align_corners=True) (backbones/hrnet.py:303-304,523-524): forward and the transposed map"""
    from iseg_amd import functional as F
    from iseg_amd import nn

    nn.set_compute_dtype(dtype)
    try:
        g = torch.Generator().manual_seed(sum(shape) + sum(size))
        x = torch.randn(shape, generator=g).to(dtype)
        dy = torch.randn((shape[0],) + tuple(size) + (shape[3],), generator=g).to(dtype)
        xg = x.cuda().requires_grad_(True)
        y = F.resize_bilinear(xg, size, align_corners=True)
        y.backward(dy.cuda())
        xr = x.double().requires_grad_(True)
        yr = O.resize_bilinear(xr, size, align_corners=True)
        yr.backward(dy.double())
        tol = 1e-5 if dtype == torch.float32 else 1.2e-2
        assert (y.detach().cpu().double() - yr.detach()).abs().max().item() <= tol * max(1.0, yr.abs().max().item())
        assert (xg.grad.cpu().double() - xr.grad).abs().max().item() <= tol * max(1.0, xr.grad.abs().max().item())
        if size[0] > 1 and size[1] > 1 and dtype == torch.float32:      # the corners map onto the corners exactly
            assert torch.equal(y[:, 0, 0].cpu(), x[:, 0, 0]) and torch.equal(y[:, -1, -1].cpu(), x[:, -1, -1])
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_resize_image_bicubic_matches_oracle(cuda, dtype):
    """utils/common.py:107-134 resize_image(method="bicubic") = tf.image.resize bicubic (half-pixel centres, Keys a = -0.5, renormalised border
    taps, 1/1024 coefficient table), up- and down-scaling on odd sizes, and its gradient (the transposed linear map)"""
    from iseg_amd.utils.common import resize_image

    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 7, 10, 16, generator=g)
    for size in ((13, 17), (5, 6), (7, 10)):
        xd = x.to(dtype).cuda().requires_grad_(True)
        y = resize_image(xd, size, method="bicubic")
        assert y.dtype == dtype and tuple(y.shape) == (2, size[0], size[1], 16)
        xr = x.to(dtype).double().requires_grad_(True)
        ref = O.resize_bicubic(xr, size)
        tol = 2e-5 if dtype == torch.float32 else 3e-2
        assert (y.detach().cpu().double() - ref.detach()).abs().max().item() < tol * max(1.0, ref.abs().max().item())
        dy = torch.randn(ref.shape, generator=g)
        y.backward(dy.to(dtype).cuda())
        ref.backward(dy.to(dtype).double())
        assert (xd.grad.cpu().double() - xr.grad).abs().max().item() < tol * max(1.0, xr.grad.abs().max().item())
    with pytest.raises(ValueError):
        resize_image(x.cuda(), (4, 4), method="lanczos3")
