"""HRNet (reference backbones/hrnet.py) through get_backbone and as a reduced network in training mode, against the fp64 restatement
(oracle/models.py hrnet_forward): transition / branch / fuse modules incl. the write-back order of the fuse module and the aligned-corner
resizes."""
import pytest
import torch

from oracle import models as OM
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a.detach().cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-8)


def _rel_l2(a, b):
    return ((a.detach().cpu().double() - b).norm() / b.norm().clamp_min(1e-12)).item()


def test_hrnet_w32_endpoints_match_oracle(cuda):
    """get_backbone("hrnet_w32"), fp32, inference statistics: the four branches and the concatenated map"""
    from iseg_amd import nn
    from iseg_amd.backbones.feature_extractor import get_backbone
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(torch.float32)
    nn.set_device("cuda:0")
    m = get_backbone("hrnet_w32", return_endpoints=True, image_shape=(1, 64, 64, 3))
    m._iseg_store = ParamStore(list(m.parameters()))
    randomize_parameters(m, 11)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 64, 96, 3, generator=g)
    with torch.no_grad():
        ends = m(x.cuda(), training=False)
    w = OM.export_weights(m)
    ref = OM.hrnet_forward(w, x.double(), [(1, [32, 64], [4, 4]), (4, [32, 64, 128], [4, 4, 4]), (3, [32, 64, 128, 256], [4, 4, 4, 4])])
    assert len(ends) == 5 and tuple(ends[-1].shape) == (1, 16, 24, 32 + 64 + 128 + 256)
    for got, want in zip(ends, ref):
        assert tuple(got.shape) == tuple(want.shape)
        assert _rel(got, want) < 5e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_small_hrnet_training_step_gradients(cuda, dtype):
    """a reduced HighResolutionNet (one block per branch, 8 / 16 / 32 channels) in training mode: outputs, input gradient and every parameter
    gradient against fp64 autograd through the restatement (batch statistics, fuse write-back, transposed aligned-corner resizes)"""
    from iseg_amd import nn
    from iseg_amd.backbones import hrnet
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        net = hrnet.HighResolutionNet(64, hrnet.Bottleneck, 4, return_endpoints=True)
        net.add_stage(1, [8, 16], hrnet.BasicBlock, [1, 1])
        net.add_stage(2, [8, 16, 32], hrnet.BasicBlock, [1, 1, 1])
        with nn.dry_run_scope():
            net(torch.empty(2, 64, 96, 3, dtype=torch.float32, device="cuda"))
        net._iseg_store = ParamStore(list(net.parameters()))
        randomize_parameters(net, 4)
        g = torch.Generator().manual_seed(9)
        x = torch.randn(2, 64, 96, 3, generator=g).to(dtype)      # (a leaf in the compute dtype: the image cast of the fp32 path carries no gradient)
        xg = x.cuda().requires_grad_(True)
        outs = net(xg, training=True)
        dys = [torch.randn(o.shape, generator=g) for o in outs]
        torch.autograd.backward([o for o in outs], [d.to(o.dtype).cuda() for o, d in zip(outs, dys)])
        w = {k: v.requires_grad_(True) if v.is_floating_point() else v for k, v in OM.export_weights(net).items()}
        xr = x.double().requires_grad_(True)
        ref = OM.hrnet_forward(w, xr, [(1, [8, 16], [1, 1]), (2, [8, 16, 32], [1, 1, 1])], training=True)
        lo = dtype == torch.bfloat16
        torch.autograd.backward(ref, [d.to(dtype).double() if lo else d.double() for d in dys])
        for got, want in zip(outs, ref):
            assert _rel(got, want.detach()) < (0.12 if lo else 2e-4)      # (bf16: ~25 layers of batch-statistics normalisation)
        missing = [p.iseg_name for p in net.parameters() if p.grad is None] + ([] if xg.grad is not None else ["dx"])
        assert not missing, missing
        if lo:
            # bf16 through ~25 batch-statistics layers with ReLU gates: element-wise gradient errors are dominated by gates that flip on
            # rounding, so the storage path is held to direction and size of the whole gradient (the fp32 run above checks every element)
            got = torch.cat([p.grad.flatten().double().cpu() for p in net.parameters()] + [xg.grad.flatten().double().cpu()])
            want = torch.cat([w[p.iseg_name].grad.flatten() for p in net.parameters()] + [xr.grad.flatten()])
            assert torch.isfinite(got).all()
            cos = float((got * want).sum() / (got.norm() * want.norm()))
            # (measured: cos 0.84, norm ratio 0.96 with 10 % forward error at the outputs; a sign or scale error would give cos < 0.5 or a ratio far from 1)
            assert cos > 0.75 and 0.8 < float(got.norm() / want.norm()) < 1.2, (cos, float(got.norm() / want.norm()))
        else:
            errs = {"dx": _rel(xg.grad, xr.grad)}
            for p in net.parameters():
                errs[p.iseg_name] = _rel(p.grad, w[p.iseg_name].grad)
            bad = {k: round(v, 4) for k, v in errs.items() if v > 3e-3}
            assert not bad, bad
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hrnet_fuse_module_forward_backward(cuda, dtype):
    """one HighResolutionFuseModule with three branches in training mode (1x1 + aligned-corner up-sampling, chains of stride-2 3x3 blocks,
    the write-back order): outputs, input gradients and parameter gradients -- shallow enough for an element-wise bf16 tolerance"""
    from iseg_amd import nn
    from iseg_amd.backbones import hrnet
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        name = "stage3/0/fuse_layers"
        mod = hrnet.HighResolutionFuseModule(True, name=name)
        shapes = [(2, 16, 24, 8), (2, 8, 12, 16), (2, 4, 6, 32)]
        with nn.dry_run_scope():
            mod([torch.empty(s, dtype=dtype, device="cuda") for s in shapes])
        mod._iseg_store = ParamStore(list(mod.parameters()))
        randomize_parameters(mod, 6)
        g = torch.Generator().manual_seed(12)
        xs = [torch.randn(s, generator=g).to(dtype) for s in shapes]
        dys = [torch.randn(s, generator=g).to(dtype) for s in shapes]
        xg = [x.cuda().requires_grad_(True) for x in xs]
        outs = mod(list(xg), training=True)
        torch.autograd.backward(outs, [d.cuda() for d in dys])
        w = {k: v.requires_grad_(True) if v.is_floating_point() else v for k, v in OM.export_weights(mod).items()}
        xr = [x.double().requires_grad_(True) for x in xs]
        ref = OM._hr_fuse(w, name, xr, True, None)
        torch.autograd.backward(ref, [d.double() for d in dys])
        lo = dtype == torch.bfloat16
        for got, want in zip(outs, ref):
            assert _rel(got, want.detach()) < (3e-2 if lo else 1e-4)
        # bf16: a ReLU gate whose pre-activation rounds across zero switches one element's gradient fully on or off (0.3-0.5 of the largest
        # gradient for a handful of elements), so the storage path is held to the L2 error of each tensor; fp32 to the largest element error
        err = _rel_l2 if lo else _rel
        errs = {f"dx{i}": err(a.grad, b.grad) for i, (a, b) in enumerate(zip(xg, xr))}
        for p in mod.parameters():
            errs[p.iseg_name] = err(p.grad, w[p.iseg_name].grad)
        bad = {k: round(v, 4) for k, v in errs.items() if v > (0.2 if lo else 2e-3)}      # (bf16 measured: <= 0.15, the gated stride-2 chain 2/0)
        assert not bad, bad
    finally:
        nn.set_compute_dtype(torch.float32)
