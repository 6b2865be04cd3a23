"""Fused ConvNeXt MLP kernels (csrc/mlp_fused.hip; reference backbones/convnext.py:51-63): the hidden [M, 4C] tile never reaches
HBM.  Checked against the fp64 restatement and against the un-fused GEMM pair on the same bf16 inputs."""
import pytest
import torch

from oracle import tf_ops as O

pytestmark = pytest.mark.gpu


def _inputs(M, C, seed, scale_w=1.0):
    g = torch.Generator().manual_seed(seed)
    y2 = torch.randn(M, C, generator=g)
    res = torch.randn(M, C, generator=g)
    W1 = torch.randn(C, 4 * C, generator=g) * (scale_w / C ** 0.5)
    W2 = torch.randn(4 * C, C, generator=g) * (scale_w / (4 * C) ** 0.5)
    b1 = torch.randn(4 * C, generator=g) * 0.3
    b2 = torch.randn(C, generator=g) * 0.3
    gamma = torch.rand(C, generator=g) + 0.5
    return y2, res, W1, b1, W2, b2, gamma


@pytest.mark.parametrize("C", [96, 192])
@pytest.mark.parametrize("M,groups", [(256, 1), (1000, 4), (4096 + 37, 3), (33, 1)])
@pytest.mark.parametrize("use_gamma,use_rs", [(True, True), (False, False)])
def test_convnext_mlp_fwd_matches_oracle(cuda, C, M, groups, use_gamma, use_rs):
    from iseg_amd import kernels as K

    y2, res, W1, b1, W2, b2, gamma = _inputs(M, C, 11 + C + M)
    bf = torch.bfloat16
    y2b, resb, W1b, W2b = (t.to(bf) for t in (y2, res, W1, W2))
    rpg = -(-M // groups)
    rs = (torch.arange(groups, dtype=torch.float32) * 0.25 + 0.5) if use_rs else None
    # the prep kernel rounds the fp32 masters to bf16 itself: hand it the already-rounded values so both sides see the same weights
    fw, _ = K.convnext_mlp_prep(W1b.float().cuda(), W2b.float().cuda(), None, backward=False)
    out = K.convnext_mlp_fwd(y2b.cuda(), fw, b1.cuda(), b2.cuda(), gamma.cuda() if use_gamma else None,
                             rs.cuda() if use_rs else None, rpg if use_rs else 0, resb.cuda())
    assert K.convnext_mlp_supported(C, bf)
    # fp64 restatement on the bf16-rounded operands (the hidden tile is rounded to bf16 before the second product, as in the kernel)
    h = y2b.double() @ W1b.double() + b1.double()
    gl = O.gelu(h).to(bf).double()
    z = gl @ W2b.double() + b2.double()
    if use_gamma:
        z = z * gamma.double()
    if use_rs:
        z = z * rs.double()[torch.arange(M) // rpg][:, None]
    ref = resb.double() + z
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-2 * max(1.0, ref.abs().max().item() / 4), err
    # same inputs through the un-fused pair of GEMMs (bf16 outputs): the two must agree to bf16 rounding
    g2 = K.dense_fwd(y2b.cuda(), W1b.cuda(), b1.cuda(), act=K.ACT_GELU)
    un = K.dense_fwd(g2, W2b.cuda(), b2.cuda(), colscale=gamma.cuda() if use_gamma else None, rowscale=rs.cuda() if use_rs else None,
                     rows_per_group=rpg if use_rs else 0, residual=resb.cuda())
    d = (out.float() - un.float()).abs().max().item()
    assert d < 6e-2, d


def test_convnext_mlp_rejects_other_shapes(cuda):
    from iseg_amd import _hip, kernels as K

    assert not K.convnext_mlp_supported(768, torch.bfloat16)      # 96-KiB weight slabs: no LDS ring fits
    assert not K.convnext_mlp_supported(384, torch.bfloat16)      # instantiated, but slower than the GEMM pair: opt-in (ISEG_MLP_FUSED_384=1)
    assert not K.convnext_mlp_supported(96, torch.float32)
    t = torch.zeros(64, 768, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(_hip.HipCallError):
        K.convnext_mlp_fwd(t, t, t, t, None, None, 0, t)
    with pytest.raises(_hip.HipCallError):
        K.convnext_mlp_bwd(t, t, t, t)


@pytest.mark.parametrize("C", [96, 192])
@pytest.mark.parametrize("M", [256, 1000, 4096 + 37, 33])
def test_convnext_mlp_bwd_chain_matches_oracle(cuda, C, M):
    """g, dh and dy2 of the recomputing backward kernel against fp64 autograd through the same bf16-rounded operands"""
    from iseg_amd import kernels as K

    y2, dbr, W1, b1, W2, b2, gamma = _inputs(M, C, 5 + C + M)
    bf = torch.bfloat16
    y2b, dbrb = y2.to(bf), dbr.to(bf)
    _, bw = K.convnext_mlp_prep(W1.cuda(), W2.cuda(), gamma.cuda(), backward=True)
    g, dh, dy2 = K.convnext_mlp_bwd(y2b.cuda(), dbrb.cuda(), bw, b1.cuda())
    W1b = W1.to(bf).double()
    W2e = (W2 * gamma).to(bf).double()
    h = (y2b.double() @ W1b + b1.double()).requires_grad_(True)
    gr = O.gelu(h)
    (dgelu,) = torch.autograd.grad(gr.sum(), h)
    dg_ref = dbrb.double() @ W2e.t()
    dh_ref = dg_ref * dgelu
    dy2_ref = dh_ref.to(bf).double() @ W1b.t()

    def rel(a, b):
        return (a.cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-8)

    assert rel(g, gr.detach()) < 1e-2, rel(g, gr.detach())
    assert rel(dh, dh_ref) < 1.5e-2, rel(dh, dh_ref)
    assert rel(dy2, dy2_ref) < 1.5e-2, rel(dy2, dy2_ref)


def test_convnext_mlp_prep_images(cuda):
    """the tiled images hold exactly the bf16-rounded kernels in the documented order (csrc/mlp_fused.hip header)"""
    from iseg_amd import kernels as K

    C = 96
    g = torch.Generator().manual_seed(3)
    W1 = torch.randn(C, 4 * C, generator=g)
    W2 = torch.randn(4 * C, C, generator=g)
    gamma = torch.rand(C, generator=g) + 0.5
    fw, bw = K.convnext_mlp_prep(W1.cuda(), W2.cuda(), gamma.cuda())
    fw, bw = fw.cpu().float(), bw.cpu().float()
    per = C * 32

    def rows_outer(t):      # a 1-KiB piece is stored [k-half][32 rows][8 k] (conflict-free ds_read_b128, round 6); the checks below read it as [32 rows][16 k]
        return t.reshape(-1, 2, 32, 8).permute(0, 2, 1, 3).reshape(t.shape)

    fw = rows_outer(fw).reshape(4 * C // 32, 2, per)
    bw = rows_outer(bw).reshape(4 * C // 32, 3, per)
    bf = torch.bfloat16
    for slab in (0, 5, 11):
        a1 = fw[slab, 0].reshape(C // 16, 32, 16)      # [kk][hid][c]
        want = W1.to(bf).float()[:, 32 * slab:32 * slab + 32].t().reshape(32, C // 16, 16).permute(1, 0, 2)
        assert torch.equal(a1, want)
        assert torch.equal(bw[slab, 0].reshape(C // 16, 32, 16), want)
        a3 = bw[slab, 1].reshape(C // 16, 32, 16)
        want3 = (W2 * gamma).to(bf).float()[32 * slab:32 * slab + 32, :].reshape(32, C // 16, 16).permute(1, 0, 2)
        assert torch.equal(a3, want3)
        # A2 / A4: [cb][s][c][kpos], kpos = 8 h + j  <->  hidden unit 16 s + 8 (j >> 2) + 4 h + (j & 3)
        kpos = torch.arange(16)
        hh, j = kpos // 8, kpos % 8
        for s_ in range(2):
            hid = 32 * slab + 16 * s_ + 8 * (j // 4) + 4 * hh + (j % 4)
            a2 = fw[slab, 1].reshape(C // 32, 2, 32, 16)[:, s_]
            want2 = W2.to(bf).float()[hid, :].t().reshape(C // 32, 32, 16)
            assert torch.equal(a2, want2)
            a4 = bw[slab, 2].reshape(C // 32, 2, 32, 16)[:, s_]
            want4 = W1.to(bf).float()[:, hid].reshape(C // 32, 32, 16)
            assert torch.equal(a4, want4)


def _bwd_reference(y2b, doutb, rs_rows, W1, b1, W2, b2, gamma):
    """fp64 restatement of the block's backward pass on the bf16-rounded operands the kernels see (hidden tile rounded to bf16 where the
    kernels round it): dy2 and every parameter gradient of backbones/convnext.py:51-57"""
    bf = torch.bfloat16
    W1b = W1.to(bf).double()
    W2e = ((W2 * gamma) if gamma is not None else W2).to(bf).double()
    dbr = doutb.double()
    if rs_rows is not None:
        dbr = (doutb.float() * rs_rows[:, None]).to(bf).double()
    h = (y2b.double() @ W1b + b1.double()).requires_grad_(True)
    gr = O.gelu(h)
    (dgelu,) = torch.autograd.grad(gr.sum(), h)
    g = gr.detach().to(bf).double()
    dh = ((dbr @ W2e.t()) * dgelu).to(bf).double()
    dy2 = dh @ W1b.t()
    Z = g.t() @ dbr
    S = dbr.sum(0)
    out = {"dy2": dy2, "dW1": y2b.double().t() @ dh, "db1": dh.sum(0)}
    if gamma is not None:
        out.update(dW2=Z * gamma.double(), dgamma=(W2.double() * Z).sum(0) + b2.double() * S, db2=gamma.double() * S)
    else:
        out.update(dW2=Z, db2=S)
    return out


@pytest.mark.parametrize("C", [96, 192])
@pytest.mark.parametrize("M,groups,use_gamma", [(256, 1, True), (1024, 4, True), (4096 + 37, 0, False), (33, 0, True), (64 * 300, 5, True)])
def test_convnext_mlp_bwd_without_hidden_tensors_matches_oracle(cuda, C, M, groups, use_gamma):
    """iseg_convnext_mlp_bwd_data + iseg_convnext_mlp_wgrad (nothing [M, 4C]-shaped in HBM) against fp64 autograd; gradients accumulate
    into their buffers; a second run reproduces the first bit for bit (fixed summation order)"""
    from iseg_amd import kernels as K

    y2, dout, W1, b1, W2, b2, gamma = _inputs(M, C, 77 + C + M)
    if not use_gamma:
        gamma = None
    bf = torch.bfloat16
    y2b, doutb = y2.to(bf), dout.to(bf)
    rpg = M // groups if groups else 0
    rs = (torch.arange(groups, dtype=torch.float32) * 0.5 + 0.25) if groups else None
    rs_rows = rs[torch.arange(M) // rpg] if groups else None
    dev = lambda t: None if t is None else t.cuda()
    _, bw = K.convnext_mlp_prep(W1.cuda(), W2.cuda(), dev(gamma), backward=True)
    ref = _bwd_reference(y2b, doutb, rs_rows, W1, b1, W2, b2, gamma)

    def run():
        grads = {k: torch.full(shape, 0.5, dtype=torch.float32, device="cuda") for k, shape in
                 (("dW1", (C, 4 * C)), ("db1", (4 * C,)), ("dW2", (4 * C, C)), ("db2", (C,)), ("dgamma", (C,)))}
        dy2 = K.convnext_mlp_bwd_data(y2b.cuda(), doutb.cuda(), bw, b1.cuda(), dev(rs), rpg)
        K.convnext_mlp_wgrad(y2b.cuda(), doutb.cuda(), bw, b1.cuda(), W2.cuda(), b2.cuda(), dev(gamma), grads["dW1"], grads["db1"], grads["dW2"],
                             grads["db2"], grads["dgamma"] if use_gamma else None, dev(rs), rpg)
        return dy2, grads

    dy2, grads = run()

    def rel(a, b):
        return (a.cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-8)

    assert rel(dy2, ref["dy2"]) < 1.5e-2, rel(dy2, ref["dy2"])
    for k in ("dW1", "db1", "dW2", "db2") + (("dgamma",) if use_gamma else ()):
        got = grads[k].cpu().double() - 0.5      # the kernels accumulate into what was there
        err = (got - ref[k]).abs().max().item() / max(ref[k].abs().max().item(), 1e-8)
        assert err < 6e-3, (k, err)
    dy2b, grads2 = run()
    assert torch.equal(dy2, dy2b)
    for k in grads:
        assert torch.equal(grads[k], grads2[k]), k


def test_convnext_mlp_wgrad_layernorm_on_load(cuda):
    """mean != NULL: the y operand is LayerNorm(y1) formed while the rows are staged -- same gradients as with y2 materialised"""
    from iseg_amd import kernels as K

    C, M = 96, 2048 + 64
    y1, dout, W1, b1, W2, b2, gamma = _inputs(M, C, 901)
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    lng, lnb = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    y1b = y1.to(bf).cuda()
    y2, mean, rstd = K.layernorm_fwd(y1b, lng.cuda(), lnb.cuda(), 1e-6)
    _, bw = K.convnext_mlp_prep(W1.cuda(), W2.cuda(), gamma.cuda(), backward=True)
    res = []
    for ln in (None, (mean, rstd, lng.cuda(), lnb.cuda())):
        grads = [torch.zeros(s, dtype=torch.float32, device="cuda") for s in ((C, 4 * C), (4 * C,), (4 * C, C), (C,), (C,))]
        K.convnext_mlp_wgrad(y2 if ln is None else y1b, dout.to(bf).cuda(), bw, b1.cuda(), W2.cuda(), b2.cuda(), gamma.cuda(), *grads, ln=ln)
        res.append(grads)
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 2e-3 * max(a.abs().max().item(), 1e-6)


@pytest.mark.parametrize("C", [96, 192])
@pytest.mark.parametrize("M", [256, 1000 + 37])
def test_convnext_mlp_layernorm_on_load_matches_the_separate_kernel(cuda, C, M):
    """iseg_convnext_mlp_fwd_ln / _bwd_data with LayerNorm folded into the row loads against LayerNorm kernel + the plain fused kernels:
    same statistics (1e-6), same outputs to bf16 rounding of y2 (the fused path rounds the same values to bf16 in registers)"""
    from iseg_amd import kernels as K

    y1, res, W1, b1, W2, b2, gamma = _inputs(M, C, 321 + C + M)
    g = torch.Generator().manual_seed(9)
    lng, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    bf = torch.bfloat16
    y1b, resb = (y1 * 1.7 + 0.3).to(bf).cuda(), res.to(bf).cuda()
    rs = torch.tensor([0.5, 1.25], device="cuda")
    rpg = -(-M // 2)
    fw, bw = K.convnext_mlp_prep(W1.cuda(), W2.cuda(), gamma.cuda(), backward=True)
    y2, mean, rstd = K.layernorm_fwd(y1b, lng, lnb, 1e-6)
    want = K.convnext_mlp_fwd(y2, fw, b1.cuda(), b2.cuda(), gamma.cuda(), rs, rpg, resb)
    got, mean2, rstd2 = K.convnext_mlp_fwd_ln(y1b, lng, lnb, 1e-6, fw, b1.cuda(), b2.cuda(), gamma.cuda(), rs, rpg, resb)
    assert (mean - mean2).abs().max().item() < 1e-5 and ((rstd - rstd2).abs() / rstd).max().item() < 1e-5
    assert (got.float() - want.float()).abs().max().item() < 4e-2 * max(1.0, want.float().abs().max().item() / 4)
    dout = torch.randn(M, C, generator=torch.Generator().manual_seed(4)).to(bf).cuda()
    want_d = K.convnext_mlp_bwd_data(y2, dout, bw, b1.cuda(), rs, rpg)
    got_d = K.convnext_mlp_bwd_data(y1b, dout, bw, b1.cuda(), rs, rpg, ln=(mean, rstd, lng, lnb))
    assert (got_d.float() - want_d.float()).abs().max().item() < 2e-2 * max(1.0, want_d.float().abs().max().item())


@pytest.mark.parametrize("C", [96, 192])
@pytest.mark.parametrize("M", [256, 1000 + 37, 4096 + 8])
def test_convnext_mlp_chain_through_the_layernorm_backward(cuda, C, M):
    """iseg_convnext_mlp_bwd_data_ln: the chain kernel's epilogue carries dy2 through the LayerNorm backward (gradient of the LayerNorm input,
    dgamma / dbeta as per-workgroup partial rows summed in fixed order) -- against the chain kernel followed by the LayerNorm backward kernel,
    and against the fp64 formula on the chain's dy2"""
    from iseg_amd import kernels as K

    y1, _, W1, b1, W2, b2, gamma = _inputs(M, C, 77 + C + M)
    g = torch.Generator().manual_seed(19)
    lng, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    bf = torch.bfloat16
    y1b = (y1 * 1.7 + 0.3).to(bf).cuda()
    rs = torch.tensor([0.5, 1.25], device="cuda")
    rpg = -(-M // 2)
    _, bw = K.convnext_mlp_prep(W1.cuda(), W2.cuda(), gamma.cuda(), backward=True)
    _, mean, rstd = K.layernorm_fwd(y1b, lng, lnb, 1e-6)
    ln = (mean, rstd, lng, lnb)
    dout = torch.randn(M, C, generator=torch.Generator().manual_seed(4)).to(bf).cuda()
    dy2 = K.convnext_mlp_bwd_data(y1b, dout, bw, b1.cuda(), rs, rpg, ln=ln)
    dg_want, db_want = torch.full((C,), 0.25, device="cuda"), torch.full((C,), -0.5, device="cuda")
    want = K.layernorm_bwd(dy2, y1b, lng, mean, rstd, dg_want, db_want)
    dg, db = torch.full((C,), 0.25, device="cuda"), torch.full((C,), -0.5, device="cuda")      # (accumulated into: the gradient buffers are live)
    got = K.convnext_mlp_bwd_data_ln(y1b, dout, bw, b1.cuda(), ln, dg, db, rs, rpg)
    # fp64 formula on the bf16-rounded dy2 the separate route hands over (the fused route keeps fp32 rows: differences are bf16 roundings of dy2)
    d, x = dy2.double().cpu(), y1b.double().cpu()
    xh = (x - mean.double().cpu()[:, None]) * rstd.double().cpu()[:, None]
    t = d * lng.double().cpu()
    ref = rstd.double().cpu()[:, None] * (t - t.mean(1, keepdim=True) - xh * (t * xh).mean(1, keepdim=True))
    scale = max(1.0, ref.abs().max().item())
    assert (got.double().cpu() - ref).abs().max().item() < 2e-2 * scale
    assert (got.float() - want.float()).abs().max().item() < 2e-2 * scale
    dg_ref, db_ref = (d * xh).sum(0) + 0.25, d.sum(0) - 0.5
    for a, b_, r in ((dg, dg_want, dg_ref), (db, db_want, db_ref)):
        tol = 5e-3 * max(1.0, r.abs().max().item())
        assert (a.double().cpu() - r).abs().max().item() < tol and (a - b_).abs().max().item() < tol
    dg2, db2 = torch.full((C,), 0.25, device="cuda"), torch.full((C,), -0.5, device="cuda")
    got2 = K.convnext_mlp_bwd_data_ln(y1b, dout, bw, b1.cuda(), ln, dg2, db2, rs, rpg)
    assert torch.equal(got, got2) and torch.equal(dg, dg2) and torch.equal(db, db2)      # no atomics anywhere: bit-reproducible
