"""One launch for all drop-path masks of a training step (reference utils/drops.py:8-22).  (Two other small fusions were measured and
dropped: drop-path gradient + column sums in one pass, and layer-scale gradients straight from the split-K slabs -- EXPERIMENTS.md.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_drop_path_pool_serves_a_step_from_one_launch(cuda):
    import iseg_amd.functional as F
    from iseg_amd import kernels as K

    keeps = [0.9, 0.8, 0.5, 1.0 - 1e-3]
    launches = []
    real_one, real_many = K.drop_path_mask, K.drop_path_masks
    K.drop_path_mask = lambda *a, **k: (launches.append("one"), real_one(*a, **k))[1]
    K.drop_path_masks = lambda *a, **k: (launches.append("many"), real_many(*a, **k))[1]
    try:
        per_step = []
        for step in range(3):
            with F.drop_path_pool():
                per_step.append([F.drop_path_factors(64, k, torch.device("cuda:0")) for k in keeps])
        assert launches == ["one"] * 4 + ["many"] + ["many"]      # the first step records, later steps draw from one launch each
        for masks in per_step:
            for m, k in zip(masks, keeps):
                assert tuple(m.shape) == (64,)
                vals = set(round(v, 5) for v in m.cpu().tolist())
                assert vals <= {0.0, round(1.0 / k, 5)}, (k, vals)      # floor(keep + u) / keep
        assert not torch.equal(per_step[1][2], per_step[2][2])          # a new draw each step
        kept = torch.stack([F.drop_path_factors(4096, 0.5, torch.device("cuda:0")) for _ in range(1)])[0]
        assert 0.4 < (kept > 0).float().mean().item() < 0.6
        with F.drop_path_pool():                                          # another batch size: records its own plan, the first one stays
            a = F.drop_path_factors(32, 0.9, torch.device("cuda:0"))
        assert tuple(a.shape) == (32,)
        # plans are kept per sample count (round 4): switching back and forth costs no extra launches and no extra seeds -- the random stream of a
        # step does not depend on the batch shapes of the steps before it (a replayed graph freezes the pooled form)
        del launches[:]
        for n in (64, 32, 64, 32):
            with F.drop_path_pool():
                got = [F.drop_path_factors(n, k, torch.device("cuda:0")) for k in (keeps if n == 64 else [0.9])]
            assert all(tuple(m.shape) == (n,) for m in got)
        assert launches == ["many"] * 4, launches
    finally:
        K.drop_path_mask, K.drop_path_masks = real_one, real_many


def test_transposed_kernel_copies_follow_the_weights(cuda):
    """nn.wt(): K-contiguous bf16 copies of 2-D kernels, all refreshed by one iseg_transpose_batched launch after a weight update"""
    from iseg_amd import kernels as K
    from iseg_amd import nn
    from iseg_amd.optimizers.modern import SGD
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(torch.bfloat16)
    try:
        shapes = [(96, 384), (384, 96), (70, 24), (8, 200)]
        ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
        store = ParamStore(ps)
        for p in ps:
            t = nn.wt(p)
            assert t.shape == (p.shape[1], p.shape[0]) and t.is_contiguous()
        for p in ps:
            assert torch.equal(nn.wt(p), p.iseg_compute.t().contiguous())
        opt = SGD(learning_rate=0.5)
        opt.build(store)
        store.flat_g.fill_(1.0)
        before = nn.wt(ps[0]).clone()
        opt.apply_gradients()
        for p in ps:      # the first request after the step refreshes every registered copy
            assert torch.equal(nn.wt(p), p.iseg_compute.t().contiguous())
        assert not torch.equal(before, nn.wt(ps[0]))
        # the forward GEMM through the K-contiguous copy agrees with the [K][N] path
        x = torch.randn(300, 96, device="cuda").to(torch.bfloat16)
        a = K.dense_fwd(x, nn.w(ps[0]), None)
        b = K.dense_fwd_t(x, nn.wt(ps[0]), None)
        assert (a.float() - b.float()).abs().max().item() <= 2e-2 * a.float().abs().max().item()
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("M,Kd,N", [(1000, 768, 96), (20000, 1536, 384), (4096, 96, 384), (3000, 200, 64)])
def test_bias_gradient_on_the_weight_gradient_gemm(cuda, M, Kd, N):
    """column sums of dY from the virtual ones-row of the weight-gradient GEMM, also when K is a multiple of the 128-row tile (one more
    tile row) and under split-K"""
    from iseg_amd import kernels as K

    x = (torch.randn(M, Kd, device="cuda") * 0.5).to(torch.bfloat16)
    dy = (torch.randn(M, N, device="cuda") * 0.5).to(torch.bfloat16)
    dw = torch.full((Kd, N), 0.25, device="cuda")
    db = torch.full((N,), -1.0, device="cuda")
    K.dense_wgrad(x, dy, dw, accumulate=True, bias_grad=db)
    ref_w = 0.25 + x.double().t() @ dy.double()
    ref_b = -1.0 + dy.double().sum(0)
    assert (dw.double() - ref_w).abs().max().item() <= 2e-3 * ref_w.abs().max().item()
    assert (db.double() - ref_b).abs().max().item() <= 2e-3 * ref_b.abs().max().item()
    dw2 = torch.empty_like(dw)
    db2 = torch.empty_like(db)
    K.dense_wgrad(x, dy, dw2, accumulate=False, bias_grad=db2)
    assert (dw2.double() - (ref_w - 0.25)).abs().max().item() <= 2e-3 * ref_w.abs().max().item()
    assert (db2.double() - (ref_b + 1.0)).abs().max().item() <= 2e-3 * ref_b.abs().max().item()


def test_deferred_gradient_reductions_match_immediate_ones(cuda):
    """K.deferred_reductions: LayerNorm / depthwise / bias gradient reductions queued and run by one launch give the same gradients (same
    summation order inside and across workgroups: BIT-identical), also
    when two reductions hit the same gradient (conflict -> early flush) and when the arena is small"""
    from iseg_amd import kernels as K

    torch.manual_seed(0)
    flat = torch.zeros(200000, device="cuda")
    views = {"lng": flat[0:384], "lnb": flat[512:896], "dw": flat[1024:1024 + 49 * 96], "db": flat[8192:8192 + 96], "cb": flat[9000:9000 + 192],
             "lng2": flat[10240:10240 + 96], "lnb2": flat[10496:10496 + 96]}
    x = torch.randn(3000, 384, device="cuda").to(torch.bfloat16)
    dy = torch.randn(3000, 384, device="cuda").to(torch.bfloat16)
    g = torch.rand(384, device="cuda") + 0.5
    b = torch.zeros(384, device="cuda")
    _, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6)
    xs = torch.randn(4096, 96, device="cuda").to(torch.bfloat16)
    dys = torch.randn(4096, 96, device="cuda").to(torch.bfloat16)
    gs = torch.rand(96, device="cuda") + 0.5
    _, means, rstds = K.layernorm_fwd(xs, gs, torch.zeros(96, device="cuda"), 1e-6)
    xi = torch.randn(2, 32, 32, 96, device="cuda").to(torch.bfloat16)
    di = torch.randn(2, 32, 32, 96, device="cuda").to(torch.bfloat16)
    cs = torch.randn(5000, 192, device="cuda").to(torch.bfloat16)
    outside = torch.zeros(192, device="cuda")

    def run():
        flat.zero_()
        outside.zero_()
        K.layernorm_bwd(dy, x, g, mean, rstd, views["lng"], views["lnb"])
        K.dwconv2d_bwd_weight(xi, di, views["dw"].view(49, 96), views["db"], 7, 1, 3, 3)
        K.colsum(cs, 192, 0, 1, 5000, 192, views["cb"], accumulate=True)
        K.colsum(cs, 192, 0, 1, 5000, 192, outside, accumulate=True)          # not a gradient: always immediate
        K.layernorm_bwd(dys, xs, gs, means, rstds, views["lng2"], views["lnb2"])
        K.layernorm_bwd(dys, xs, gs, means, rstds, views["lng2"], views["lnb2"])    # same outputs again: conflict with the queued one
        K.colsum(cs, 192, 0, 1, 5000, 192, views["cb"], accumulate=True)
        return flat.clone(), outside.clone()

    ref, ref_out = run()
    for arena in (96 << 20, 1 << 20):
        with K.deferred_reductions(flat, arena_bytes=arena):
            got, got_out = run()         # (flat.clone() inside reads the buffer before the flush: compare after the exit instead)
        assert torch.equal(flat, ref), (flat - ref).abs().max().item()
        assert torch.equal(outside, ref_out)
    assert not K._DEFER["active"]


def test_layerscale_grads_from_split_k_slabs(cuda):
    """iseg_layerscale_grads_slabs sums the weight-gradient product's split-K slabs (and their ones-row S) while it reads them: same dW2 / dgamma /
    db2 as slab sum -> Z -> iseg_layerscale_grads, and both against the fp64 formulas of backbones/convnext.py:56-57"""
    from iseg_amd import kernels as K

    torch.manual_seed(3)
    M, Kd, Nd = 16384, 1536, 384
    g = (torch.randn(M, Kd) * 0.5).to(torch.bfloat16).cuda()
    dbr = (torch.randn(M, Nd) * 0.1).to(torch.bfloat16).cuda()
    W2, b2, gamma = torch.randn(Kd, Nd).cuda() * 0.05, torch.randn(Nd).cuda() * 0.1, (torch.rand(Nd) + 0.5).cuda()
    sl = K.dense_wgrad_slabs(g, dbr)
    assert sl is not None and sl[1] > 1, "the flagship's stage-2 product must be split"
    outs = []
    for route in ("slabs", "tensor"):
        dW2, dg, db = torch.full((Kd, Nd), 0.5, device="cuda"), torch.full((Nd,), -1.0, device="cuda"), torch.full((Nd,), 2.0, device="cuda")
        if route == "slabs":
            K.layerscale_grads_slabs(sl[0], sl[1], W2, b2, gamma, dW2, dg, db)
        else:
            Z, S = torch.empty(Kd, Nd, device="cuda"), torch.empty(Nd, device="cuda")
            K.dense_wgrad(g, dbr, Z, accumulate=False, bias_grad=S)
            K.layerscale_grads(Z, W2, b2, gamma, S, dW2, dg, db)
        outs.append((dW2, dg, db))
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
    Zr = g.double().cpu().t() @ dbr.double().cpu()
    Sr = dbr.double().cpu().sum(0)
    want = (Zr * gamma.double().cpu() + 0.5, (W2.double().cpu() * Zr).sum(0) + b2.double().cpu() * Sr - 1.0, gamma.double().cpu() * Sr + 2.0)
    for a, r in zip(outs[0], want):
        assert (a.double().cpu() - r).abs().max().item() <= 2e-4 * max(1.0, r.abs().max().item())
