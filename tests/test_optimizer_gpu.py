"""Optimizer step kernels (csrc/optim.hip; reference optimizers/modern/adamw.py:13-74, optimizers/modern/sgd.py:12-51, Keras' base
optimizer clipping as driven by core_optimizer.py:170-183) against the oracle at Keras' default epsilon 1e-7: NaN gradients,
clipvalue / clipnorm (per variable) / global_clipnorm, per-variable lr_multiplier, variables excluded from weight decay, amsgrad,
nesterov, the l2 regulariser -- several steps over a flat buffer of odd-sized variables."""
import math

import pytest
import torch

from oracle import tf_ops as O

pytestmark = pytest.mark.gpu

SHAPES = [(7, 5), (300,), (3, 3, 8, 16), (1,), (257,), (64, 33)]
NAMES = ["a/kernel", "a/bias", "b/kernel", "c/gamma", "b/beta", "d/kernel"]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    ps = []
    for i, (shp, name) in enumerate(zip(SHAPES, NAMES)):
        p = torch.nn.Parameter(torch.randn(shp, generator=g).cuda())
        p.iseg_name = name
        ps.append(p)
    ps[2].lr_multiplier = 10.0      # utils/train_utils.py:75-87 set_weights_lr_multiplier
    ps[4].lr_multiplier = 0.1
    return ps


def _grads(step, seed, nan=True):
    g = torch.Generator().manual_seed(1000 * seed + step)
    gs = [torch.randn(shp, generator=g) * (3.0 if i == 0 else 0.3) for i, shp in enumerate(SHAPES)]
    if nan:
        gs[1][5] = float("nan")
        gs[2][0, 1, 2, 3] = float("nan")
    return gs


CLIPS = [dict(), dict(clipvalue=0.25), dict(clipnorm=1.0), dict(global_clipnorm=2.0), dict(clipnorm=1e6)]


@pytest.mark.parametrize("clip", CLIPS, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()) or "noclip")
@pytest.mark.parametrize("amsgrad", [False, True])
def test_adamw_step_matches_oracle(cuda, clip, amsgrad):
    from iseg_amd.optimizers.modern import AdamW
    from iseg_amd.param_store import ParamStore

    ps = _params(0)
    store = ParamStore(ps)
    lr_fn = lambda it: 1e-2 * (1.0 - it / 10.0)      # noqa: E731  a schedule: evaluated at `iterations`
    opt = AdamW(learning_rate=lr_fn, weight_decay=0.05, amsgrad=amsgrad, **clip)      # epsilon stays at Keras' 1e-7
    opt.exclude_from_weight_decay(var_names=["bias", "gamma", "beta"])
    opt.build(store)
    opt.grad_scale = 0.5      # 1 / world of a two-replica job: the summed gradient is averaged before clipping
    w = [p.data.detach().cpu().double() for p in ps]
    m = [torch.zeros_like(x) for x in w]
    v = [torch.zeros_like(x) for x in w]
    vh = [torch.zeros_like(x) for x in w]
    for step in range(4):
        gs = _grads(step, 3)
        for p, g in zip(ps, gs):
            p.grad.copy_(g.cuda())
        opt.apply_gradients()
        lr = lr_fn(step)
        gd = O.clip_gradients([O.scrub_nan(g.double()) * 0.5 for g in gs], **clip)
        for i, p in enumerate(ps):
            wd = 0.0 if any(k in NAMES[i] for k in ("bias", "gamma", "beta")) else 0.05
            mult = float(getattr(p, "lr_multiplier", 1.0))
            if amsgrad:
                w[i], m[i], v[i], vh[i] = O.adamw_step(w[i], gd[i], m[i], v[i], step + 1, lr, mult, wd, vhat=vh[i])
            else:
                w[i], m[i], v[i] = O.adamw_step(w[i], gd[i], m[i], v[i], step + 1, lr, mult, wd)
        for i, (p, o, n) in enumerate(store.segments):
            got = p.data.cpu().double()
            assert torch.isfinite(got).all()
            err = (got - w[i]).abs().max().item()
            assert err <= 2e-6 + 2e-4 * lr * float(getattr(p, "lr_multiplier", 1.0)), (step, NAMES[i], err)
            assert (opt.m[o:o + n].view(p.shape).cpu().double() - m[i]).abs().max().item() <= 1e-6 * max(1.0, m[i].abs().max().item())
            assert (opt.v[o:o + n].view(p.shape).cpu().double() - v[i]).abs().max().item() <= 1e-6 * max(1.0, v[i].abs().max().item())
            if amsgrad:
                assert (opt.vhat[o:o + n].view(p.shape).cpu().double() - vh[i]).abs().max().item() <= 1e-6 * max(1.0, vh[i].abs().max().item())
            # the bf16 shadow the MFMA kernels read is the rounded master
            assert torch.equal(p.iseg_compute.cpu(), p.data.cpu().to(torch.bfloat16))
    assert opt.iterations == 4


@pytest.mark.parametrize("clip", CLIPS[:4], ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()) or "noclip")
@pytest.mark.parametrize("nesterov", [False, True])
def test_sgd_momentum_step_matches_oracle(cuda, clip, nesterov):
    from iseg_amd.optimizers.modern import SGD
    from iseg_amd.param_store import ParamStore

    ps = _params(1)
    ps[0].l2_regularizer = 1e-2      # utils/keras_ops.py set_weight_decay
    ps[5].l2_regularizer = 5e-3
    store = ParamStore(ps)
    opt = SGD(learning_rate=0.05, momentum=0.9, nesterov=nesterov, **clip)
    opt.build(store)
    w = [p.data.detach().cpu().double() for p in ps]
    m = [torch.zeros_like(x) for x in w]
    for step in range(4):
        gs = _grads(step, 5, nan=False)
        for p, g in zip(ps, gs):
            p.grad.copy_(g.cuda())
        opt.apply_gradients()
        l2 = [float(getattr(p, "l2_regularizer", 0.0)) for p in ps]
        gd = O.clip_gradients([g.double() + 2.0 * l2[i] * w[i] for i, g in enumerate(gs)], **clip)
        for i, p in enumerate(ps):
            w[i], m[i] = O.sgd_step(w[i], gd[i], m[i], 0.05, float(getattr(p, "lr_multiplier", 1.0)), 0.9, 0.0, nesterov)
        for i, (p, o, n) in enumerate(store.segments):
            assert (p.data.cpu().double() - w[i]).abs().max().item() <= 3e-6 * max(1.0, w[i].abs().max().item()), (step, NAMES[i])
            assert (opt.m[o:o + n].view(p.shape).cpu().double() - m[i]).abs().max().item() <= 3e-6 * max(1.0, m[i].abs().max().item())


def test_sgd_does_not_scrub_nan_and_adamw_does(cuda):
    """only AdamW_EXT overrides _clip_gradients (adamw.py:63-74); SGD_EXT lets a NaN gradient through (sgd.py:38-51)"""
    from iseg_amd.optimizers.modern import SGD, AdamW
    from iseg_amd.param_store import ParamStore

    for cls, poisoned in ((SGD, True), (AdamW, False)):
        ps = _params(2)
        store = ParamStore(ps)
        opt = cls(learning_rate=0.01)
        opt.build(store)
        for p, g in zip(ps, _grads(0, 7, nan=True)):
            p.grad.copy_(g.cuda())
        opt.apply_gradients()
        assert bool(torch.isnan(ps[1].data[5])) == poisoned
        assert not torch.isnan(ps[0].data).any()


def test_only_one_clip_option(cuda):
    from iseg_amd.optimizers.modern import AdamW

    with pytest.raises(ValueError):
        AdamW(clipnorm=1.0, clipvalue=0.5)


def test_clipnorm_is_deterministic(cuda):
    from iseg_amd.optimizers.modern import AdamW
    from iseg_amd.param_store import ParamStore

    outs = []
    for _ in range(2):
        ps = _params(4)
        store = ParamStore(ps)
        opt = AdamW(learning_rate=0.01, global_clipnorm=0.5)
        opt.build(store)
        for p, g in zip(ps, _grads(1, 9)):
            p.grad.copy_(g.cuda())
        opt.apply_gradients()
        outs.append(store.flat_w.clone())
    assert torch.equal(outs[0], outs[1])
    assert math.isfinite(outs[0].sum().item())
