"""ConvNeXt V2 (reference backbones/convnext_v2.py): the Global Response Normalization operator (csrc/grn.hip) through the C ABI, the V2 block
and the whole backbone as get_backbone builds it, against the fp64 restatement (oracle/tf_ops.py grn, oracle/models.py convnext_v2_*)."""
import pytest
import torch

from oracle import models as OM
from oracle import tf_ops as O
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a.detach().cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-8)


# C/8 below, at and above the 256 chunk lanes of a workgroup; one-pixel planes; more samples than pixels; ragged row counts
GRN_SHAPES = [(2, 5, 7, 320), (3, 8, 8, 384), (1, 1, 1, 64), (2, 33, 17, 2560), (17, 3, 3, 8), (2, 16, 16, 2048), (1, 40, 40, 1280)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", GRN_SHAPES)
def test_grn_forward_backward_match_oracle(cuda, dtype, shape):
    from iseg_amd import kernels as K

    N, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(dtype)
    dy = torch.randn(shape, generator=g).to(dtype)
    gamma = torch.randn(C, generator=g) * 0.5
    beta = torch.randn(C, generator=g) * 0.1
    xc, dyc = x.cuda().reshape(N, H * W, C), dy.cuda().reshape(N, H * W, C)
    y, nx, gx = K.grn_fwd(xc, gamma.cuda(), beta.cuda(), 1e-6)
    dgamma = torch.full((C,), 1.0, device="cuda")      # (+)= semantics: the starting value must survive
    dbeta = torch.full((C,), -2.0, device="cuda")
    dx = K.grn_bwd(dyc, xc, gamma.cuda(), nx, gx, dgamma, dbeta, 1e-6)

    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = O.grn(xr, gr, br, 1e-6)
    yr.backward(dy.double())
    gxr = torch.sqrt((x.double() ** 2).sum(dim=(1, 2)) + 1e-6)
    assert _rel(gx, gxr) < 1e-5
    assert _rel(nx, gxr / (gxr.mean(dim=-1, keepdim=True) + 1e-6)) < 1e-5
    lo = dtype == torch.bfloat16
    assert _rel(y.reshape(shape), yr.detach()) < (1e-2 if lo else 2e-6)
    assert _rel(dx.reshape(shape), xr.grad) < (1e-2 if lo else 1e-5)
    assert _rel(dgamma - 1.0, gr.grad) < 2e-5 + (1e-6 if not lo else 0)
    assert _rel(dbeta + 2.0, br.grad) < 2e-5


def test_grn_is_reproducible_and_zero_gamma_is_identity(cuda):
    from iseg_amd import kernels as K

    torch.manual_seed(5)
    x = torch.randn(4, 24 * 24, 1280, device="cuda").to(torch.bfloat16)
    dy = torch.randn_like(x)
    gamma = torch.randn(1280, device="cuda")
    beta = torch.randn(1280, device="cuda") * 0.1

    def run():
        y, nx, gx = K.grn_fwd(x, gamma, beta, 1e-6)
        dg, db = torch.zeros(1280, device="cuda"), torch.zeros(1280, device="cuda")
        dx = K.grn_bwd(dy, x, gamma, nx, gx, dg, db, 1e-6)
        return y, nx, dx, dg, db

    a, b = run(), run()
    for u, v in zip(a, b):
        assert torch.equal(u, v)      # fixed summation order everywhere
    # the layer's initial state (gamma = beta = 0, convnext_v2.py:29-41) passes activations and gradients through untouched
    z = torch.zeros(1280, device="cuda")
    y, nx, gx = K.grn_fwd(x, z, z, 1e-6)
    assert torch.equal(y, x)
    dg, db = torch.zeros(1280, device="cuda"), torch.zeros(1280, device="cuda")
    dx = K.grn_bwd(dy, x, z, nx, gx, dg, db, 1e-6)
    assert torch.equal(dx, dy)
    assert dg.abs().max().item() > 0      # ... while gamma itself still learns


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grn_backward_folds_the_activation_derivative(cuda, dtype):
    """mul: dx of iseg_grn_bwd times a saved derivative tensor == the two steps done separately (rounded once instead of twice)"""
    from iseg_amd import kernels as K

    torch.manual_seed(9)
    x = torch.randn(3, 50, 384, device="cuda").to(dtype)
    dy = torch.randn_like(x)
    mul = torch.rand_like(x) + 0.25
    gamma = torch.randn(384, device="cuda")
    _, nx, gx = K.grn_fwd(x, gamma, gamma, 1e-6)
    dg0, db0 = torch.zeros(384, device="cuda"), torch.zeros(384, device="cuda")
    dg1, db1 = torch.zeros(384, device="cuda"), torch.zeros(384, device="cuda")
    plain = K.grn_bwd(dy, x, gamma, nx, gx, dg0, db0, 1e-6)
    folded = K.grn_bwd(dy, x, gamma, nx, gx, dg1, db1, 1e-6, mul=mul)
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)      # parameter gradients do not see the multiplier
    want = plain.double() * mul.double()
    tol = 1e-6 if dtype == torch.float32 else 1.2e-2
    assert ((folded.double() - want).abs() <= tol * want.abs() + tol).all()
    with pytest.raises(ValueError):
        K.grn_bwd(dy, x, gamma, nx, gx, dg1, db1, 1e-6, mul=mul[:, :, :8])


def test_grn_rejects_bad_arguments(cuda):
    from iseg_amd import _hip, kernels as K

    x = torch.zeros(2, 9, 20, dtype=torch.bfloat16, device="cuda")      # C % 8 != 0
    z = torch.zeros(20, device="cuda")
    with pytest.raises(_hip.HipCallError):
        K.grn_fwd(x, z, z, 1e-6)
    assert _hip.lib().iseg_grn_workspace_bytes(0, 9, 16) == 0
    x = torch.zeros(2, 9, 16, dtype=torch.bfloat16, device="cuda")
    z = torch.zeros(16, device="cuda")
    L = _hip.lib()
    rc = L.iseg_grn_fwd(K.ptr(x), K.ptr(z), K.ptr(z), K.ptr(x), K.ptr(z), K.ptr(z), 2, 9, 16, 1e-6, K.BF16, K.ptr(z), 16, K.stream())
    assert rc != 0      # workspace too small


def test_grn_parameter_gradients_through_the_deferred_queue(cuda):
    """inside K.deferred_reductions the per-sample rows of (dgamma | dbeta) go to the trainer's arena and are summed by the batched launch"""
    from iseg_amd import kernels as K

    torch.manual_seed(2)
    C = 640
    flat = torch.zeros(8192, device="cuda")
    dg, db = flat[256:256 + C], flat[1024:1024 + C]
    x = torch.randn(6, 100, C, device="cuda").to(torch.bfloat16)
    dy = torch.randn_like(x)
    gamma = torch.randn(C, device="cuda")
    y, nx, gx = K.grn_fwd(x, gamma, gamma, 1e-6)
    dx_ref = K.grn_bwd(dy, x, gamma, nx, gx, dg, db, 1e-6)
    ref = flat.clone()
    flat.zero_()
    with K.deferred_reductions(flat, arena_bytes=1 << 20):
        dx = K.grn_bwd(dy, x, gamma, nx, gx, dg, db, 1e-6)
    assert torch.equal(dx, dx_ref)
    assert torch.equal(flat, ref)      # same rows, same fixed order


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,dil,dp", [((2, 8, 8, 80), 1, 0.0), ((3, 9, 7, 160), 2, 0.2), ((2, 4, 4, 320), 1, 0.5), ((2, 2, 2, 640), 1, 0.1)])
def test_convnext_v2_block_forward_backward(cuda, monkeypatch, dtype, shape, dil, dp, fused):
    """the block as one tape node (default) and layer by layer through the generic operators (ISEG_V2_BLOCK_FUSED=0)"""
    from iseg_amd import nn
    from iseg_amd.backbones.convnext_v2 import Block
    from iseg_amd.param_store import ParamStore

    monkeypatch.setenv("ISEG_V2_BLOCK_FUSED", fused)
    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        N, H, W, C = shape
        blk = Block(C, drop_path_prob=dp, name="stages/0/0")
        blk.dwconv.dilation_rate = (dil, dil)
        with nn.dry_run_scope():
            blk(torch.empty(shape, dtype=dtype, device="cuda"))
        assert tuple(blk.grn.gamma.shape) == (1, 1, 1, 4 * C) and float(blk.grn.gamma.abs().max()) == 0.0
        store = ParamStore(list(blk.parameters()))
        blk._iseg_store = store
        randomize_parameters(blk, 3)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(shape, generator=g).to(dtype)
        dy = torch.randn(shape, generator=g).to(dtype)
        f = None
        if dp > 0:
            keep = 1 - dp
            f = torch.floor(keep + torch.rand(N, generator=g)) / keep
            f[0] = 1 / keep
            blk.drop_path_mask = f.float().cuda()
        xg = x.cuda().requires_grad_(True)
        y = blk(xg, training=True)
        y.backward(dy.cuda())
        w = {k: v.requires_grad_(True) for k, v in OM.export_weights(blk).items()}
        xr = x.double().requires_grad_(True)
        yr = OM.convnext_v2_block(w, "stages/0/0", xr, dil, None if f is None else f.double())
        yr.backward(dy.double())
        tol = 2e-4 if dtype == torch.float32 else 4e-2
        assert _rel(y, yr.detach()) < (1e-5 if dtype == torch.float32 else 2e-2)
        errs = {"dx": _rel(xg.grad, xr.grad)}
        for p in blk.parameters():
            errs[p.iseg_name] = _rel(p.grad, w[p.iseg_name].grad)
        bad = {k: v for k, v in errs.items() if v > tol}
        assert not bad, bad
    finally:
        nn.set_compute_dtype(torch.float32)


def test_grn_fold_kernels_match_the_unfolded_operator(cuda):
    """the normalisation folded into the next Dense: scaled kernel copies + bias reproduce grn(g) @ W + b through iseg_gemm's per-row-group B,
    and the folded backward (per-sample g^T dbr products) reproduces dW, dgamma, dbeta and the data gradient of the plain operators"""
    from iseg_amd import kernels as K

    torch.manual_seed(3)
    N, HW, C4, Co = 3, 512, 256, 64
    bf = torch.bfloat16
    g = torch.randn(N, HW, C4, device="cuda").to(bf)
    W = torch.randn(C4, Co, device="cuda") * C4 ** -0.5
    b = torch.randn(Co, device="cuda") * 0.1
    gamma = torch.randn(C4, device="cuda") * 0.5
    beta = torch.randn(C4, device="cuda") * 0.1
    wt = W.to(bf).t().contiguous()
    nx, gx = K.grn_stats(g, 1e-6)
    z, nx2, gx2 = K.grn_fwd(g, gamma, beta, 1e-6)
    assert torch.equal(nx, nx2) and torch.equal(gx, gx2)
    # forward
    w2n = K.grn_fold_weights(wt, gamma, nx)
    a = gamma * nx + 1.0
    assert torch.equal(w2n, (wt.float()[None] * a[:, None, :]).to(bf))
    bias2 = K.grn_fold_bias(W, beta, b)
    assert torch.allclose(bias2, b + beta @ W, rtol=1e-5, atol=1e-5)
    M = N * HW
    res = torch.randn(M, Co, device="cuda").to(bf)
    out = torch.empty(M, Co, dtype=bf, device="cuda")
    K.gemm(g.reshape(M, C4), w2n, out, M, Co, C4, lda=C4, ldb=C4, ldd=Co, a_kcontig=1, b_kcontig=1, bias=bias2, residual=res, ldr=Co,
           b_group=(HW, Co * C4))
    want = (g.double() * a.double()[:, None, :] + beta.double()).reshape(M, C4) @ W.double() + b.double() + res.double()
    assert _rel(out, want.cpu()) < 1e-2
    # the unsupported layouts are refused, not mis-computed
    from iseg_amd import _hip
    with pytest.raises(_hip.HipCallError):
        K.gemm(g.reshape(M, C4), w2n, out, M, Co, C4, lda=C4, ldb=C4, ldd=Co, a_kcontig=1, b_kcontig=1, b_group=(HW - 8, Co * C4))
    # backward
    dbr = torch.randn(M, Co, device="cuda").to(bf)
    d = (torch.rand(M, C4, device="cuda") + 0.25).to(bf)
    dW_ref = torch.zeros(C4, Co, device="cuda")
    db_ref = torch.zeros(Co, device="cuda")
    K.dense_wgrad(z.reshape(M, C4), dbr, dW_ref, bias_grad=db_ref)
    dz = K.dense_dgrad(dbr, W.to(bf))
    dg_ref, dbt_ref = torch.zeros(C4, device="cuda"), torch.zeros(C4, device="cuda")
    dh_ref = K.grn_bwd(dz.reshape(N, HW, C4), g, gamma, nx, gx, dg_ref, dbt_ref, 1e-6, mul=d)
    for sps in (1, 2):
        rows = HW // sps
        slabs = torch.empty(N * sps, C4, Co, device="cuda")
        K.gemm(g.reshape(M, C4), dbr, slabs, C4, Co, rows, lda=C4, ldb=Co, ldd=Co, a_kcontig=0, b_kcontig=0, batch=N * sps, batch_inner=1,
               sa=(rows * C4, 0), sb=(rows * Co, 0), sd=(C4 * Co, 0))
        S = K.colsum(dbr, Co, 0, 1, M, Co, torch.empty(Co, device="cuda"))
        dW = torch.zeros(C4, Co, device="cuda")
        dstats = K.grn_fold_wgrad(slabs, sps, W, gamma, beta, nx, S, dW)
        # (the reference product used the bf16-rounded grn(g); the folded one scales exact per-sample products)
        assert _rel(dW, dW_ref.cpu().double()) < 6e-3
        assert torch.allclose(S, db_ref, rtol=1e-4, atol=1e-3)
        dg, dbt = torch.zeros(C4, device="cuda"), torch.zeros(C4, device="cuda")
        dh = K.grn_bwd_folded(dz.reshape(N, HW, C4), g, gamma, nx, gx, dstats, dg, dbt, 1e-6, mul=d)
        assert _rel(dg, dg_ref.cpu().double()) < 6e-3 and _rel(dbt, dbt_ref.cpu().double()) < 6e-3
        assert _rel(dh, dh_ref.cpu().double()) < 1.5e-2


@pytest.mark.parametrize("shape,dp", [((2, 32, 32, 64), 0.0), ((2, 32, 32, 96), 0.3)])
def test_convnext_v2_block_with_the_normalisation_folded(cuda, monkeypatch, shape, dp):
    """wide planes take the folded route (functional._ConvNeXtV2BlockFn): it must agree with the oracle and with the unfolded route"""
    from iseg_amd import nn
    from iseg_amd.backbones.convnext_v2 import Block
    from iseg_amd.param_store import ParamStore

    dtype = torch.bfloat16
    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        N, H, W, C = shape
        outs = {}
        for fold in ("1", "0"):
            monkeypatch.setenv("ISEG_V2_GRN_FOLD", fold)
            blk = Block(C, drop_path_prob=dp, name="stages/0/0")
            with nn.dry_run_scope():
                blk(torch.empty(shape, dtype=dtype, device="cuda"))
            blk._iseg_store = ParamStore(list(blk.parameters()))
            randomize_parameters(blk, 3)
            g = torch.Generator().manual_seed(1)
            x = torch.randn(shape, generator=g).to(dtype)
            dy = torch.randn(shape, generator=g).to(dtype)
            f = None
            if dp > 0:
                keep = 1 - dp
                f = torch.floor(keep + torch.rand(N, generator=g)) / keep
                f[0] = 1 / keep
                blk.drop_path_mask = f.float().cuda()
            xg = x.cuda().requires_grad_(True)
            y = blk(xg, training=True)
            assert y.grad_fn.fold == (fold == "1")
            y.backward(dy.cuda())
            outs[fold] = (y.detach(), xg.grad, {p.iseg_name: p.grad.clone() for p in blk.parameters()})
            if fold == "1":
                w = {k: v.requires_grad_(True) for k, v in OM.export_weights(blk).items()}
                xr = x.double().requires_grad_(True)
                yr = OM.convnext_v2_block(w, "stages/0/0", xr, 1, None if f is None else f.double())
                yr.backward(dy.double())
                assert _rel(y, yr.detach()) < 2e-2
                errs = {"dx": _rel(xg.grad, xr.grad)}
                for p in blk.parameters():
                    errs[p.iseg_name] = _rel(p.grad, w[p.iseg_name].grad)
                bad = {k: v for k, v in errs.items() if v > 4e-2}
                assert not bad, bad
        ya, dxa, ga = outs["1"]
        yb, dxb, gb = outs["0"]
        assert _rel(ya, yb.cpu().double()) < 2e-2 and _rel(dxa, dxb.cpu().double()) < 3e-2
        for k in ga:
            assert _rel(ga[k], gb[k].cpu().double()) < 3e-2, k
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("output_stride", [32, 8])
def test_convnext_v2_nano_endpoints_match_oracle(cuda, output_stride):
    """get_backbone("convnext_v2_nano") with the dilation surgery, fp32, all four endpoints and the input gradient"""
    from iseg_amd import nn
    from iseg_amd.backbones.feature_extractor import get_backbone
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(torch.float32)
    nn.set_device("cuda:0")
    m = get_backbone("convnext_v2_nano", output_stride=output_stride, return_endpoints=True, image_shape=(1, 64, 64, 3))
    m._iseg_store = ParamStore(list(m.parameters()))
    randomize_parameters(m, 7)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 64, 64, 3, generator=g)
    ends = m(x.cuda(), training=False)
    w = OM.export_weights(m)
    ref = OM.convnext_v2_backbone(w, x.double(), depths=(2, 2, 8, 2), output_stride=output_stride)
    assert len(ends) == 5 and ends[0] is None
    for got, want in zip(ends[1:], ref[1:]):
        assert tuple(got.shape) == tuple(want.shape)
        assert _rel(got, want) < 2e-4
