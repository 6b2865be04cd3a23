"""DCNv3 sampling core (csrc/dcnv3.hip) against the op-for-op transcription of layers/dcn_v3/{op,utils}.py in the oracle -- the
[y,x] / [x,y] quirk and the clipped-corner weights included -- and a small InternImage built from it."""
import pytest
import torch

from oracle import models as OM
from oracle import tf_ops as O
from tests.test_kernels_gpu import DTYPES, close, q, rnd
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,G,scale", [((2, 9, 7, 32), 2, 1.0), ((1, 12, 12, 64), 4, 1.0), ((2, 5, 8, 24), 3, 2.0), ((1, 6, 6, 12), 4, 1.0)])
def test_dcnv3_core_forward_backward(cuda, dtype, shape, G, scale):
    from iseg_amd import functional as F
    from iseg_amd import nn

    nn.set_compute_dtype(dtype)
    try:
        N, H, W, C = shape
        Cg = C // G
        x, xr = q(rnd(shape, 1), dtype)
        off, offr = q(rnd((N, H, W, G * 9 * 2), 2) * 1.5, dtype)       # offsets of +-1.5 pixels: samples cross cell and image borders
        m, mr = q(torch.softmax(rnd((N, H, W, G, 9), 3), -1).reshape(N, H, W, G * 9), dtype)
        for t in (x, off, m, xr, offr, mr):
            t.requires_grad_(True)
        y = F.dcnv3_core(x, off, m, G, Cg, (3, 3), 1, 1, 1, scale)
        yr = O.dcnv3_op(xr, offr, mr, (3, 3), (1, 1), "SAME", (1, 1), G, Cg, scale)
        assert tuple(y.shape) == tuple(yr.shape)
        close(y, yr, dtype, "dcnv3 fwd", f32_tol=1e-5, bf16_tol=1.5e-2)
        dy, dyr = q(rnd(shape, 4), dtype)
        y.backward(dy)
        yr.backward(dyr)
        close(x.grad, xr.grad, dtype, "dcnv3 dx", f32_tol=2e-5, bf16_tol=2e-2)
        close(m.grad, mr.grad, dtype, "dcnv3 dmask", f32_tol=2e-5, bf16_tol=2e-2)
        if dtype == torch.float32:
            close(off.grad, offr.grad, dtype, "dcnv3 doffset", f32_tol=5e-5)
        else:
            # bf16 offsets are multiples of 2^-8 or coarser: some sampling coordinates land exactly on a cell border, where
            # floor() in fp32 (kernel) and fp64 (oracle) may pick different cells and the offset gradient is discontinuous;
            # bound the relative L2 error instead of the max-norm
            d = off.grad.cpu().double() - offr.grad
            assert d.norm().item() / offr.grad.norm().item() < 5e-2
    finally:
        nn.set_compute_dtype(torch.float32)


def test_dcnv3_zero_offsets_reproduce_the_reference_base_grid(cuda):
    """with zero offsets and a one-hot mask on the centre tap the op must return the reference's (transposed-grid) resampling,
    not the identity: pins the [y,x] quirk independently of random data"""
    from iseg_amd import kernels as K

    N, H, W, G, Cg = 1, 6, 6, 1, 8
    x = torch.arange(H * W, dtype=torch.float32).reshape(1, H, W, 1).repeat(1, 1, 1, Cg)
    off = torch.zeros(N, H, W, 18)
    m = torch.zeros(N, H, W, 9)
    m[..., 4] = 1.0
    y = K.dcnv3_fwd(x.cuda(), off.cuda(), m.cuda(), G, Cg, 3, 3, 1, 1, 1, 1.0).cpu()
    yr = O.dcnv3_op(x.double(), off.double(), m.double(), (3, 3), (1, 1), "SAME", (1, 1), G, Cg, 1.0)
    assert (y.double() - yr).abs().max().item() < 1e-4
    assert (y - x).abs().max().item() > 1.0      # NOT the identity


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("post_norm,cfs", [(False, False), (True, False), (True, True)])
def test_intern_image_small_member(cuda, dtype, post_norm, cfs):
    """cfs: use_center_feature_scale (intern_image_huge: post-norm + centre-feature scale, layers/dcn_v3/dcn_v3.py:138-146)"""
    from iseg_amd import nn
    from iseg_amd.backbones.intern_image.intern_image import InternImage
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        shape = (2, 40, 56, 3)
        net = InternImage(stem_filters=32, depths=[1, 2], groups=[2, 4], drop_path_rate=0.2, layer_scale=1.0, use_post_norm=post_norm,
                          use_center_feature_scale=cfs, return_endpoints=True, name="ii_test")
        with nn.dry_run_scope():
            net(torch.empty(shape, dtype=torch.float32, device="cuda"))
        net._iseg_store = ParamStore(list(net.parameters()))
        randomize_parameters(net, 9)
        g = torch.Generator().manual_seed(0)
        with torch.no_grad():       # offsets of about +-1 pixel so that sampling is genuinely deformed (zeros-initialised by default)
            for p in net.parameters():
                if p.iseg_name.endswith("offset/bias"):
                    p.copy_(torch.randn(p.shape, generator=g).to(p.device))
        net._iseg_store.sync_shadow()
        fa, fb = torch.tensor([1.25, 0.0]), torch.tensor([1.25, 1.25])
        dp = [[None], [(fa.double(), fb.double()), (fb.double(), fa.double())]]
        for bi in range(2):
            for li, f in enumerate(dp[bi]):
                if f is not None:
                    net.blocks[bi].blocks[li].drop_path_masks = tuple(t.float().cuda() for t in f)
        x = torch.randn(shape, generator=g)
        eps = net(x.cuda(), training=True)
        w = {k_: v.requires_grad_(True) for k_, v in OM.export_weights(net).items()}
        ref = OM.intern_image_forward(w, x.double(), (1, 2), (2, 4), post_norm, dp)
        assert [tuple(e.shape) for e in eps] == [tuple(r.shape) for r in ref]

        def rel(a, b):
            d = a.detach().cpu().double() - b
            return d.norm().item() / max(b.norm().item(), 1e-8) if a.dtype == torch.bfloat16 else d.abs().max().item() / b.abs().max().item()

        for a, b in zip(eps, ref):
            assert rel(a, b.detach()) < (2e-4 if dtype == torch.float32 else 4e-2)
        dys = [torch.randn(tuple(r.shape), generator=g).to(dtype) for r in ref]
        torch.autograd.backward(list(eps), [d.cuda() for d in dys])
        torch.autograd.backward(ref, [d.double() for d in dys])
        gmax = max(v.grad.abs().max().item() for v in w.values() if v.grad is not None)
        bad = {}
        for p in net.parameters():
            r = w[p.iseg_name].grad
            d = p.grad.detach().cpu().double() - r
            if dtype == torch.float32:
                e = d.abs().max().item() / max(r.abs().max().item(), 1e-3 * gmax)
            else:
                e = d.norm().item() / max(r.norm().item(), 1e-3 * gmax * r.numel() ** 0.5)
            if e > (1e-3 if dtype == torch.float32 else 0.2):   # bf16: see the border remark in the core test
                bad[p.iseg_name] = e
        assert not bad, bad
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,G,spread", [((1, 40, 37, 32), 2, 1.5), ((2, 33, 30, 16), 2, 9.0), ((1, 44, 50, 48), 3, 4.0)])
def test_dcnv3_backward_windows_and_side_buffer(cuda, dtype, shape, G, spread):
    """several 16 x 16 output tiles (windows overlap, the gather sums them in tile order) and offsets far beyond the window margin (the
    int64 side buffer): same gradients as the oracle, and bit-identical from run to run"""
    from iseg_amd import _hip
    from iseg_amd import kernels as K

    N, H, W, C = shape
    Cg = C // G
    assert _hip.lib().iseg_dcnv3_bwd_workspace_bytes(N, H, W, G, Cg, 3, 3, 1, 1, 1, 1.0) > 0      # the deterministic path serves these shapes
    x, xr = q(rnd(shape, 11), dtype)
    off, offr = q(rnd((N, H, W, G * 9 * 2), 12) * spread, dtype)
    m, mr = q(torch.softmax(rnd((N, H, W, G, 9), 13), -1).reshape(N, H, W, G * 9), dtype)
    dy, dyr = q(rnd(shape, 14), dtype)
    for t in (xr, offr, mr):
        t.requires_grad_(True)
    O.dcnv3_op(xr, offr, mr, (3, 3), (1, 1), "SAME", (1, 1), G, Cg, 1.0).backward(dyr)
    dx, doff, dm = K.dcnv3_bwd(x, off, m, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    close(dx, xr.grad, torch.float32, "dcnv3 dx", f32_tol=2e-5 if dtype == torch.float32 else 2e-2)
    close(dm, mr.grad, dtype, "dcnv3 dmask", f32_tol=2e-5, bf16_tol=2e-2)
    dx2, doff2, dm2 = K.dcnv3_bwd(x, off, m, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert torch.equal(dx, dx2) and torch.equal(doff, doff2) and torch.equal(dm, dm2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,G", [((2, 21, 19, 24), 3), ((1, 17, 23, 12), 4), ((1, 30, 26, 8), 2)])
def test_dcnv3_general_backward_is_bit_identical_and_right(cuda, dtype, shape, G):
    """the window route (8 per group) and the group widths it does not serve (4 per group: a lane per channel; 3: a lane per group): the input gradient is
    accumulated by int64 fixed-point atomics (round 4: no float atomics), so repeated calls give the same bits; values against the oracle"""
    from iseg_amd import kernels as K

    N, H, W, C = shape
    Cg = C // G
    x, xr = q(rnd(shape, 21), dtype)
    off, offr = q(rnd((N, H, W, G * 9 * 2), 22) * 2.0, dtype)
    m, mr = q(torch.softmax(rnd((N, H, W, G, 9), 23), -1).reshape(N, H, W, G * 9), dtype)
    dy, dyr = q(rnd(shape, 24), dtype)
    for t in (xr, offr, mr):
        t.requires_grad_(True)
    O.dcnv3_op(xr, offr, mr, (3, 3), (1, 1), "SAME", (1, 1), G, Cg, 1.0).backward(dyr)
    outs = [K.dcnv3_bwd(x, off, m, dy, G, Cg, 3, 3, 1, 1, 1, 1.0) for _ in range(3)]
    close(outs[0][0], xr.grad, torch.float32, "dcnv3 dx", f32_tol=2e-5 if dtype == torch.float32 else 2e-2)
    close(outs[0][2], mr.grad, dtype, "dcnv3 dmask", f32_tol=2e-5, bf16_tol=2e-2)
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,G,spread", [((1, 40, 37, 32), 2, 1.5), ((1, 30, 26, 8), 2, 9.0), ((1, 17, 23, 12), 4, 2.0)])
def test_dcnv3_input_gradient_keeps_its_precision_for_tiny_gradients(cuda, dtype, shape, G, spread):
    """output gradients of 1e-8 -- the order a mean loss over 512 x 512 pixels hands a first-stage layer -- on the window route and on the general
    route (4- and 3-wide groups), side buffer included: single contributions are ~3e-10 and the 2^-40 = 9.1e-13 accumulators must resolve them.
    (Rounds 4-5 converted the SIGNED value: the fraction of a small negative number was rounded to fp32 next to 1, a resolution of 2.3e-10 -- every
    negative contribution of this test off by tens of per cent; found by the 512 x 512 whole-model parity of round 6.)"""
    from iseg_amd import kernels as K

    N, H, W, C = shape
    Cg = C // G
    x, xr = q(rnd(shape, 31), dtype)
    off, offr = q(rnd((N, H, W, G * 9 * 2), 32) * spread, dtype)
    m, mr = q(torch.softmax(rnd((N, H, W, G, 9), 33), -1).reshape(N, H, W, G * 9), dtype)
    dy, dyr = q(rnd(shape, 34) * 1e-8, dtype)
    xr.requires_grad_(True)
    O.dcnv3_op(xr, offr, mr, (3, 3), (1, 1), "SAME", (1, 1), G, Cg, 1.0).backward(dyr)
    dx, _, _ = K.dcnv3_bwd(x, off, m, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    want = xr.grad
    err = (dx.double().cpu() - want).norm().item() / want.norm().item()
    assert err < (5e-3 if dtype == torch.float32 else 1.5e-2), err


# ---- DCNv2 / FaPN (layers/dcn_v2.py, layers/fapn.py; round 5) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H,W,C,Fo,custom", [(2, 9, 7, 16, 24, True), (1, 12, 12, 32, 32, False), (2, 5, 6, 8, 16, True)])
def test_dcnv2_layer_forward_and_gradients(cuda, dtype, N, H, W, C, Fo, custom):
    """DCNv2 (modulated deformable sampling kernels + offset convolution + one GEMM) against the op-for-op restatement of layers/dcn_v2.py:110-262:
    offsets large enough to leave the image (the clipped-corner weights and the zero border take part), output, input / offset-input gradients and
    every parameter gradient.  bf16: offsets land on cell borders where the offset gradient is discontinuous -- direction / size band only."""
    from iseg_amd import nn
    from iseg_amd.layers.dcn_v2 import DCNv2
    from iseg_amd.param_store import ParamStore
    from oracle import models as OM
    from oracle import tf_ops as O
    from tests.util_models import randomize_parameters

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        g = torch.Generator().manual_seed(3)
        x = torch.randn((N, H, W, C), generator=g).to(dtype)
        o = (torch.randn((N, H, W, C), generator=g) * 1.5).to(dtype)
        layer = DCNv2(Fo, (3, 3), use_custom_offset=custom, activation="relu", name="dcn")
        ins = [torch.empty((N, H, W, C), dtype=dtype, device="cuda")] * 2 if custom else torch.empty((N, H, W, C), dtype=dtype, device="cuda")
        with nn.dry_run_scope():
            layer(ins)
        layer._iseg_store = ParamStore(list(layer.parameters()))
        randomize_parameters(layer, 5)
        xg, og = x.cuda().requires_grad_(True), o.cuda().requires_grad_(True)
        y = layer([xg, og]) if custom else layer(xg)
        w = {k: v.requires_grad_(True) for k, v in OM.export_weights(layer).items()}
        xr, orr = x.double().requires_grad_(True), o.double().requires_grad_(True)
        yr = torch.relu(O.dcnv2(xr, orr if custom else xr, w["dcn/kernel"], w["dcn/bias"], w["dcn/offset_kernel"], w["dcn/offset_bias"]))
        f32 = dtype == torch.float32
        scale = yr.abs().max().item()
        assert (y.detach().cpu().double() - yr.detach()).abs().max().item() < (2e-5 if f32 else 4e-2) * scale
        dy = torch.randn(tuple(yr.shape), generator=g).to(dtype)
        y.backward(dy.cuda())
        yr.backward(dy.double())

        def rel(a, b):
            return (a.detach().cpu().double() - b).norm().item() / max(b.norm().item(), 1e-12)

        tol = 2e-4 if f32 else 0.25
        assert rel(xg.grad, xr.grad) < tol
        if custom:
            assert rel(og.grad, orr.grad) < tol
        for p in layer.parameters():
            assert rel(p.grad, w[p.iseg_name].grad) < tol, p.iseg_name
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fapn_decoder_matches_oracle(cuda, dtype):
    """FeatureAlignedPyramidNet (layers/fapn.py:83-140) on three pyramid levels: squeeze-and-excitation gate + projection of the skip, bilinear
    up-sampling of the coarser level, offsets from both, DCNv2 alignment, relu, sum -- every level and (fp32) every gradient"""
    from iseg_amd import nn
    from iseg_amd.layers.fapn import FeatureAlignedPyramidNet
    from iseg_amd.param_store import ParamStore
    from oracle import models as OM
    from tests.util_models import randomize_parameters

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        shapes = [(2, 16, 20, 24), (2, 8, 10, 40), (2, 4, 5, 32)]      # the coarsest level already has skip_conv_filters channels
        fapn = FeatureAlignedPyramidNet(skip_conv_filters=32, name="fapn")
        with nn.dry_run_scope():
            fapn([torch.empty(s, dtype=dtype, device="cuda") for s in shapes])
        fapn._iseg_store = ParamStore(list(fapn.parameters()))
        randomize_parameters(fapn, 9)
        g = torch.Generator().manual_seed(1)
        xs = [torch.randn(s, generator=g).to(dtype) for s in shapes]
        xg = [t.cuda().requires_grad_(True) for t in xs]
        outs = fapn(xg, training=False)
        w = {k: v.requires_grad_(True) for k, v in OM.export_weights(fapn).items()}
        xr = [t.double().requires_grad_(True) for t in xs]
        ref = OM.fapn_forward(w, "fapn", xr)
        assert len(outs) == len(ref) == 3 and tuple(outs[0].shape) == (2, 16, 20, 32)
        f32 = dtype == torch.float32
        for a, b in zip(outs, ref):
            assert (a.detach().cpu().double() - b.detach()).abs().max().item() < (5e-5 if f32 else 6e-2) * b.abs().max().item()
        dy = torch.randn(tuple(ref[0].shape), generator=g).to(dtype)
        outs[0].backward(dy.cuda())
        ref[0].backward(dy.double())
        if f32:
            for a, b in zip(xg, xr):
                assert (a.grad.cpu().double() - b.grad).norm().item() < 5e-4 * b.grad.norm().item()
            for p in fapn.parameters():
                r = w[p.iseg_name].grad
                assert (p.grad.cpu().double() - r).norm().item() < 1e-3 * max(r.norm().item(), 1e-9), p.iseg_name
        else:
            assert all(torch.isfinite(a.grad.float()).all() for a in xg)
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("C,G,shape", [(112, 7, (2, 20, 24)), (64, 4, (1, 33, 17)), (224, 14, (1, 16, 16))])
def test_dcnv3_layer_joint_projection_matches_layerwise(cuda, monkeypatch, C, G, shape):
    """bf16 storage: the offset | mask projections as ONE product with the sampling kernels reading its column ranges (F._DcnJointFn, round 5)
    against the layer-by-layer route (two Dense layers, softmax_groups, dcnv3_core) on the same weights: output, input gradient and every parameter
    gradient (offset / mask kernels and biases included) to bf16 rounding of the intermediate tensors"""
    from iseg_amd import functional as F
    from iseg_amd import nn
    from iseg_amd.layers.dcn_v3.dcn_v3 import DeformableConvolutionV3
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(torch.bfloat16)
    nn.set_device("cuda:0")
    try:
        N, H, W = shape
        layer = DeformableConvolutionV3(filters=C, groups=G, name="dcn_joint_test")
        with nn.dry_run_scope():
            layer(torch.empty((N, H, W, C), dtype=torch.bfloat16, device="cuda"))
        store = ParamStore(list(layer.parameters()))
        randomize_parameters(layer, 3)
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for p in layer.parameters():
                if "offset" in p.iseg_name or "mask" in p.iseg_name:      # zeros-initialised by default: make the sampling genuinely deformed
                    p.copy_((torch.randn(p.shape, generator=g) * (0.5 if p.dim() == 1 else 0.05)).to(p.device))
        store.sync_shadow()
        x = torch.randn((N, H, W, C), generator=g).bfloat16()
        dy = torch.randn((N, H, W, C), generator=g).bfloat16()

        def run(joint):
            monkeypatch.setattr(F, "_DCN_JOINT", joint)
            store.zero_grad()
            xg = x.cuda().requires_grad_(True)
            y = layer(xg, training=True)
            y.backward(dy.cuda())
            return y.detach().float().cpu(), xg.grad.float().cpu(), {p.iseg_name: p.grad.detach().float().cpu().clone() for p in layer.parameters()}

        ya, dxa, ga = run(True)
        yb, dxb, gb = run(False)

        def rel(a, b):
            return (a - b).norm().item() / max(b.norm().item(), 1e-12)

        assert rel(ya, yb) < 1e-2, rel(ya, yb)
        assert rel(dxa, dxb) < 2e-2, rel(dxa, dxb)
        bad = {k: rel(ga[k], gb[k]) for k in ga if rel(ga[k], gb[k]) > 3e-2}
        assert not bad, bad
        # a weight update: the joint weight images are per-update derivations (nn.joint_kernels, refreshed with the other prepared images) -- stale
        # images would reproduce the OLD projection
        with torch.no_grad():
            for p in layer.parameters():
                if "offset" in p.iseg_name or "mask" in p.iseg_name:
                    p.add_((torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 1 else 0.03)).to(p.device))
        store.sync_shadow()
        yc, dxc, gc = run(True)
        yd, dxd, gd = run(False)
        assert rel(yc, ya) > 5e-2, "the update did not change the output: the test is vacuous"
        assert rel(yc, yd) < 1e-2 and rel(dxc, dxd) < 2e-2, (rel(yc, yd), rel(dxc, dxd))
        bad = {k: rel(gc[k], gd[k]) for k in gc if rel(gc[k], gd[k]) > 3e-2}
        assert not bad, bad
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,G,spread", [((1, 40, 37, 32), 2, 1.5), ((2, 33, 30, 16), 2, 9.0), ((1, 44, 50, 48), 3, 4.0)])
def test_dcnv3_joint_layout_backward_and_kept_side_buffer(cuda, dtype, shape, G, spread):
    """iseg_dcnv3_fwd_ld / _bwd_ld (round 5): offsets and mask as column ranges of one [pixels, ld] matrix, dx in the storage type, the side buffer
    kept between calls instead of zeroed per call -- with offsets far beyond the window margin, so that the side buffer IS used: the same values
    as the dense entry points (forward bit-identical; dx to the storage rounding), bit-identical from call to call, and the kept buffer all zero
    again afterwards"""
    from iseg_amd import kernels as K

    N, H, W, C = shape
    Cg = C // G
    gp = G * 9
    ld = (3 * gp + 7) // 8 * 8 + 8      # (one more chunk of padding than the projection would leave)
    x, _ = q(rnd(shape, 11), dtype)
    off, _ = q(rnd((N, H, W, 2 * gp), 12) * spread, dtype)
    m, _ = q(torch.softmax(rnd((N, H, W, G, 9), 13), -1).reshape(N, H, W, gp), dtype)
    dy, _ = q(rnd(shape, 14), dtype)
    om = torch.full((N * H * W, ld), float("nan"), dtype=dtype, device="cuda")
    om[:, :2 * gp] = off.reshape(-1, 2 * gp)
    om[:, 2 * gp:3 * gp] = m.reshape(-1, gp)
    y = K.dcnv3_fwd(x, off, m, G, Cg, 3, 3, 1, 1, 1, 1.0)
    yj = K.dcnv3_fwd_joint(x, om, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert torch.equal(y, yj)
    dx, doff, dm = K.dcnv3_bwd(x, off, m, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    dxj, dom = K.dcnv3_bwd_joint(x, om, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert dxj.dtype == dtype
    assert torch.equal(dom[:, :2 * gp], doff.reshape(-1, 2 * gp)) and torch.equal(dom[:, 2 * gp:3 * gp], dm.reshape(-1, gp))
    assert torch.equal(dxj, dx.to(dtype))
    dxj2, dom2 = K.dcnv3_bwd_joint(x, om, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert torch.equal(dxj, dxj2) and torch.equal(dom[:, :3 * gp], dom2[:, :3 * gp])
    side = K._DCN_SIDE[str(x.device)]
    assert int(side.view(torch.int32).ne(0).sum().item()) == 0, "the kept side buffer was not left all zero"


@pytest.mark.parametrize("dtype", DTYPES)
def test_dcnv3_backward_reports_non_finite_gradients(cuda, dtype):
    """the integer window accumulators would turn an inf / NaN arriving gradient into finite garbage: the window kernel raises a flag instead and the
    gather kernel writes NaN for the input gradient (both entry points); the next call with finite operands is clean again (the kept flag word is
    reset)"""
    from iseg_amd import kernels as K

    shape, G = (1, 40, 37, 32), 2
    N, H, W, C = shape
    Cg, gp = C // G, G * 9
    x, _ = q(rnd(shape, 11), dtype)
    off, _ = q(rnd((N, H, W, 2 * gp), 12), dtype)
    m, _ = q(torch.softmax(rnd((N, H, W, G, 9), 13), -1).reshape(N, H, W, gp), dtype)
    dy, _ = q(rnd(shape, 14), dtype)
    bad = dy.clone()
    bad[0, 7, 9, 3] = float("inf")
    dx, _, _ = K.dcnv3_bwd(x, off, m, bad, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert bool(torch.isnan(dx).all())
    ld = (3 * gp + 7) // 8 * 8
    om = torch.zeros((N * H * W, ld), dtype=dtype, device="cuda")
    om[:, :2 * gp] = off.reshape(-1, 2 * gp)
    om[:, 2 * gp:3 * gp] = m.reshape(-1, gp)
    dxj, _ = K.dcnv3_bwd_joint(x, om, bad, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert bool(torch.isnan(dxj).all())
    dxj, _ = K.dcnv3_bwd_joint(x, om, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert bool(torch.isfinite(dxj).all())
    dx, _, _ = K.dcnv3_bwd(x, off, m, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert torch.equal(dxj, dx.to(dtype))


def test_dcnv3_joint_backward_with_a_column_sliced_matrix_and_after_an_unfinished_call(cuda):
    """round-5 advisor: (1) an offset | mask matrix handed over as a column slice (shape[1] < row pitch) gets its gradient matrix at the SAME pitch
    (the C ABI takes one pitch pair for both); (2) a backward call that did not finish leaves the kept side buffer in an unknown state: the
    host-side dirty bit makes the next call zero-fill it first, so stale fixed-point partials or a set poison flag cannot reach later gradients"""
    from iseg_amd import kernels as K

    dtype, shape, G, spread = torch.bfloat16, (2, 33, 30, 16), 2, 9.0
    N, H, W, C = shape
    Cg, gp = C // G, G * 9
    ld = (3 * gp + 7) // 8 * 8 + 8
    x, _ = q(rnd(shape, 11), dtype)
    off, _ = q(rnd((N, H, W, 2 * gp), 12) * spread, dtype)
    m, _ = q(torch.softmax(rnd((N, H, W, G, 9), 13), -1).reshape(N, H, W, gp), dtype)
    dy, _ = q(rnd(shape, 14), dtype)
    full = torch.zeros((N * H * W, ld), dtype=dtype, device="cuda")
    full[:, :2 * gp] = off.reshape(-1, 2 * gp)
    full[:, 2 * gp:3 * gp] = m.reshape(-1, gp)
    dx0, dom0 = K.dcnv3_bwd_joint(x, full, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    sliced = full[:, :3 * gp]
    assert sliced.stride(0) == ld and sliced.shape[1] == 3 * gp
    dx1, dom1 = K.dcnv3_bwd_joint(x, sliced, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert dom1.stride(0) == ld and tuple(dom1.shape) == tuple(sliced.shape)
    assert torch.equal(dx0, dx1) and torch.equal(dom0[:, :3 * gp], dom1)
    # an "unfinished" call: the guard is still set and the buffer holds garbage
    key = str(x.device)
    side = K._DCN_SIDE[key]
    side.view(torch.int32).fill_(0x7F7F7F7F)
    K._dcn_side_guard(x.device, True)
    dx2, dom2 = K.dcnv3_bwd_joint(x, full, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert torch.equal(dx0, dx2) and torch.equal(dom0[:, :3 * gp], dom2[:, :3 * gp])
    assert key not in K._DCN_SIDE_DIRTY
    assert int(K._DCN_SIDE[key].view(torch.int32).ne(0).sum().item()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,G", [(8, 2), (24, 2), (24, 1), (64, 2)])      # Cg = 4 (channel-lane kernel), 12 (scalar), 24 (8-wide), 32 (channel-lane)
def test_dcnv3_general_backward_route_reports_non_finite_gradients(cuda, dtype, C, G):
    """round-5 verdict item 5(d) / advisor: the general route (group widths other than 8 / 16) accumulates the input gradient in int64 fixed point
    too, and its conversion saturated an inf / NaN contribution to a finite value: the accumulating kernels now raise a flag word in the workspace and
    the conversion pass writes NaN for the input gradient, as the window route does; a following call with finite operands is clean"""
    from iseg_amd import kernels as K

    N, H, W = 1, 20, 18
    Cg, gp = C // G, G * 9
    x, _ = q(rnd((N, H, W, C), 11), dtype)
    off, _ = q(rnd((N, H, W, 2 * gp), 12), dtype)
    m, _ = q(torch.softmax(rnd((N, H, W, G, 9), 13), -1).reshape(N, H, W, gp), dtype)
    dy, _ = q(rnd((N, H, W, C), 14), dtype)
    clean, _, _ = K.dcnv3_bwd(x, off, m, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert bool(torch.isfinite(clean).all())
    for poison in (float("inf"), float("nan")):
        bad = dy.clone()
        bad[0, 7, 9, 3] = poison
        dx, _, _ = K.dcnv3_bwd(x, off, m, bad, G, Cg, 3, 3, 1, 1, 1, 1.0)
        assert bool(torch.isnan(dx).all()), poison
    again, _, _ = K.dcnv3_bwd(x, off, m, dy, G, Cg, 3, 3, 1, 1, 1, 1.0)
    assert torch.equal(clean, again)
