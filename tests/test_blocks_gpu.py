"""The fused ConvNeXt block operator (backbones/convnext.py:47-63 of the reference) in isolation: forward and every gradient vs
the oracle, with layer scale, injected drop-path factors and dilation."""
import pytest
import torch

from oracle import models as OM
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,dil,dp", [((3, 2, 2, 768), 1, 0.1), ((2, 8, 8, 96), 1, 0.0), ((2, 9, 7, 192), 2, 0.2), ((4, 4, 4, 384), 1, 0.5)])
def test_convnext_block_forward_backward(cuda, dtype, shape, dil, dp):
    from iseg_amd import nn
    from iseg_amd.backbones.convnext import Block
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        N, H, W, C = shape
        blk = Block(C, drop_path_prob=dp, layer_scale_init_value=1.0, name="stages/0/0")
        blk.dwconv.dilation_rate = (dil, dil)
        with nn.dry_run_scope():
            blk(torch.empty(shape, dtype=dtype, device="cuda"))
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
        yr = OM.convnext_block(w, "stages/0/0", xr, dil, None if f is None else f.double())
        yr.backward(dy.double())
        tol = 2e-4 if dtype == torch.float32 else 4e-2

        def rel(a, b):
            return (a.detach().cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-8)

        assert rel(y, yr.detach()) < (1e-5 if dtype == torch.float32 else 2e-2)
        errs = {"dx": rel(xg.grad, xr.grad)}
        for p in blk.parameters():
            errs[p.iseg_name] = rel(p.grad, w[p.iseg_name].grad)
        bad = {k: v for k, v in errs.items() if v > tol}
        assert not bad, bad
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("shape", [(4, 32, 32, 384), (3, 16, 16, 768)])
def test_convnext_block_with_the_drop_path_factor_folded_into_the_saved_activation(cuda, monkeypatch, shape):
    """ISEG_DP_FOLDED=1 (round 6, off by default: measured equal): x + s gamma (g W2 + b2) = x + gamma ((s g) W2 + s b2) -- the pwconv1 epilogue saves
    s g, pwconv2 scales its bias row-wise, the backward pass works on the unscaled dout (row factor in the x-aux epilogue, S = colsum(s dout) inside the
    layer-scale launch).  Same block, same inputs, both routes at the un-fused stages' widths: outputs and every gradient agree to bf16 rounding, and
    both agree with the oracle"""
    from iseg_amd import functional as F
    from iseg_amd import nn
    from iseg_amd.backbones.convnext import Block
    from iseg_amd.param_store import ParamStore

    dtype = torch.bfloat16
    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        N, H, W, C = shape
        g = torch.Generator().manual_seed(1)
        x = torch.randn(shape, generator=g).to(dtype)
        dy = torch.randn(shape, generator=g).to(dtype)
        f = torch.tensor([1.25, 0.0, 1.25, 1.25][:N])

        def run(folded):
            monkeypatch.setattr(F, "_DP_FOLDED", folded)
            blk = Block(C, drop_path_prob=0.2, layer_scale_init_value=1.0, name="stages/2/0")
            with nn.dry_run_scope():
                blk(torch.empty(shape, dtype=dtype, device="cuda"))
            blk._iseg_store = ParamStore(list(blk.parameters()))
            randomize_parameters(blk, 3)
            blk.drop_path_mask = f.float().cuda()
            xg = x.cuda().requires_grad_(True)
            y = blk(xg, training=True)
            y.backward(dy.cuda())
            from iseg_amd import kernels as K

            K.deferred_flush()
            torch.cuda.synchronize()
            return blk, y.detach(), xg.grad.detach(), {p.iseg_name: p.grad.detach().clone() for p in blk.parameters()}

        blk, y0, dx0, g0 = run(False)
        _, y1, dx1, g1 = run(True)
        w = {k: v.requires_grad_(True) for k, v in OM.export_weights(blk).items()}
        xr = x.double().requires_grad_(True)
        yr = OM.convnext_block(w, "stages/2/0", xr, 1, f.double())
        yr.backward(dy.double())

        def rel(a, b):
            return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item() / max(b.detach().abs().max().item(), 1e-8)

        assert rel(y1, yr) < 2e-2 and rel(y1, y0) < 2e-2
        assert torch.equal(y1[1], x.cuda()[1])      # the dropped sample passes through untouched on both routes
        assert rel(dx1, xr.grad) < 4e-2
        bad = {k: rel(g1[k], w[k].grad) for k in g1 if rel(g1[k], w[k].grad) > 4e-2}
        assert not bad, bad
    finally:
        nn.set_compute_dtype(torch.float32)
