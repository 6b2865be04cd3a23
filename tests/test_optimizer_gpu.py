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


class _Holder(torch.nn.Module):
    """stands in for a Keras layer: MultiOptimizer.create_optimizer_spec only reads the variables' names"""

    def __init__(self, params):
        super().__init__()
        for i, p in enumerate(params):
            self.register_parameter(f"p{i}", p)


def test_multi_optimizer_routes_variables_by_layer(cuda):
    """optimizers/multi_optimizer.py:10-63: AdamW owns the variables of the first layer group, SGD-momentum those of the second; each
    variable follows ITS optimizer's oracle update (schedule, multiplier, decay exclusions, clipvalue) and is left alone by the other; a
    variable claimed twice is refused; a list of optimizers reaches the trainer through model.multi_optimizers_layers()"""
    from iseg_amd.optimizers.modern import SGD, AdamW
    from iseg_amd.optimizers.multi_optimizer import MultiOptimizer
    from iseg_amd.param_store import ParamStore

    ps = _params(4)
    store = ParamStore(ps)
    ga, gb = _Holder(ps[:3]), _Holder(ps[3:])
    adam = AdamW(learning_rate=lambda it: 1e-2 / (1 + it), weight_decay=0.05, clipvalue=0.5)
    sgd = SGD(learning_rate=0.05, momentum=0.9)
    mo = MultiOptimizer(optimizers_and_layers=[(adam, [ga]), (sgd, gb)])
    assert mo.optimizer_specs[0]["weights"] == NAMES[:3] and mo.optimizer_specs[1]["weights"] == NAMES[3:]
    mo.exclude_from_weight_decay(var_names=["bias", "gamma", "beta"])
    mo.build(store)
    w = [p.data.detach().cpu().double() for p in ps]
    m = [torch.zeros_like(x) for x in w]
    v = [torch.zeros_like(x) for x in w]
    for step in range(3):
        gs = _grads(step, 11, nan=False)
        for p, g in zip(ps, gs):
            p.grad.copy_(g.cuda())
        mo.apply_gradients()
        for i, p in enumerate(ps):
            mult = float(getattr(p, "lr_multiplier", 1.0))
            if i < 3:
                wd = 0.0 if any(k in NAMES[i] for k in ("bias", "gamma", "beta")) else 0.05
                g = O.clip_gradients([gs[i].double()], clipvalue=0.5)[0]
                w[i], m[i], v[i] = O.adamw_step(w[i], g, m[i], v[i], step + 1, 1e-2 / (1 + step), mult, wd)
            else:
                w[i], m[i] = O.sgd_step(w[i], gs[i].double(), m[i], 0.05, mult, 0.9, 0.0, False)
            assert (p.data.cpu().double() - w[i]).abs().max().item() <= 3e-6 * max(1.0, w[i].abs().max().item()), (step, NAMES[i])
            assert torch.equal(p.iseg_compute.cpu(), p.data.cpu().to(torch.bfloat16))
    assert mo.iterations == 3 and adam.iterations == 3 and sgd.iterations == 3
    with pytest.raises(ValueError):
        MultiOptimizer(optimizers_and_layers=[(AdamW(), [ga]), (SGD(), [ga, gb])]).build(store)
    with pytest.raises(NotImplementedError):
        MultiOptimizer(optimizers_and_layers=[(AdamW(global_clipnorm=1.0), [ga]), (SGD(), [gb])]).build(store)


def test_multi_optimizer_nan_gradient_stays_in_its_group(cuda):
    """a NaN gradient on a variable owned by the AdamW group is scrubbed there (adamw.py:63-74) and must not reach the weight through the
    SGD group's pass (learning-rate multiplier 0 for variables it does not own: NaN * 0 = NaN if the kernel computed them)"""
    from iseg_amd.optimizers.modern import SGD, AdamW
    from iseg_amd.optimizers.multi_optimizer import MultiOptimizer
    from iseg_amd.param_store import ParamStore

    ps = _params(4)
    store = ParamStore(ps)
    mo = MultiOptimizer(optimizers_and_layers=[(AdamW(learning_rate=1e-2, weight_decay=0.0), [_Holder(ps[:3])]),
                                               (SGD(learning_rate=0.05, momentum=0.9), _Holder(ps[3:]))])
    mo.build(store)
    before = [p.data.clone() for p in ps]
    gs = _grads(0, 11, nan=False)
    for p, g in zip(ps, gs):
        p.grad.copy_(g.cuda())
    ps[0].grad.view(-1)[3] = float("nan")      # (only NaN is scrubbed: adamw.py:70 tf.where(is_nan(g), 0, g))
    ps[1].grad.view(-1)[0] = float("nan")
    mo.apply_gradients()
    for p in ps:
        assert torch.isfinite(p.data).all() and torch.isfinite(p.iseg_compute.float()).all(), p.iseg_name
    sgd = mo.optimizer_specs[1]["optimizer"]
    assert torch.isfinite(sgd.m).all()
    # the scrubbed element behaves like a zero gradient: no movement on the first AdamW step without decay
    assert ps[0].data.view(-1)[3] == before[0].view(-1)[3]
    assert not torch.equal(ps[3].data, before[3])


def test_trainer_takes_a_list_of_optimizers_through_multi_optimizers_layers(cuda):
    from iseg_amd import heads, nn
    from iseg_amd.data import synthetic_batch
    from iseg_amd.optimizers.modern import SGD, AdamW
    from iseg_amd.optimizers.multi_optimizer import MultiOptimizer
    from iseg_amd.trainer import TrainableModel

    nn.set_compute_dtype(torch.bfloat16)
    nn.set_device("cuda:0")
    try:
        model = heads.convnext_tiny_aspp(build_input_size=(64, 64))
        opts = [AdamW(learning_rate=1e-3, weight_decay=0.05), SGD(learning_rate=1e-2, momentum=0.9)]
        with pytest.raises(ValueError):      # the model has not named its layer groups
            TrainableModel(model, optimizer=list(opts), loss=model.custom_losses(21, 255, 2), loss_weights=model.custom_losses_weights())
        model.layers_for_multi_optimizers = [[model.backbone], [model.head, model.logits_conv]]
        tm = TrainableModel(model, optimizer=list(opts), loss=model.custom_losses(21, 255, 2), loss_weights=model.custom_losses_weights(),
                            metrics=model.custom_metrics(21, 255))
        assert isinstance(tm.optimizer, MultiOptimizer)
        owned = sum(len(s["weights"]) for s in tm.optimizer.optimizer_specs)
        assert owned == len(list(model.parameters())) + len([b for b in model.buffers() if hasattr(b, "iseg_name")])
        x, y = synthetic_batch(2, 64, 64, seed=5)
        before = {p.iseg_name: p.data.clone() for p in model.parameters()}
        losses = [float(tm.train_step(x.cuda(), y.cuda())[0].detach()) for _ in range(6)]
        assert all(l == l for l in losses) and losses[-1] < losses[0], losses
        moved = [n for n, b in before.items() if not torch.equal(b, dict((p.iseg_name, p.data) for p in model.parameters())[n])]
        assert len(moved) == len(before)      # every variable has an owner and was updated
    finally:
        nn.set_compute_dtype(torch.float32)
