"""Whole-model parity on the GPU: SegManaged(ConvNeXt-T + ASPP) forward / backward / train steps vs the CPU oracle
(fp32 storage: logits within 1e-3 abs and identical argmax masks; bf16 storage: bf16-level agreement)."""
import pytest
import torch

from oracle import models as OM
from oracle import tf_ops as O
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu


def _setup(dtype, size=(64, 64), output_stride=32, drop_path=0.0, seed=0):
    from iseg_amd import nn
    from iseg_amd.heads import convnext_tiny_aspp
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    model = convnext_tiny_aspp(num_class=21, output_stride=output_stride, build_input_size=size, drop_path_rate=drop_path,
                               dropout_rate=0.0, layer_scale_init_value=1.0)
    model._iseg_store = ParamStore(list(model.parameters()))
    randomize_parameters(model, seed)
    return model


@pytest.fixture(autouse=True)
def _restore_policy():
    from iseg_amd import nn

    yield
    nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("size,os_", [((64, 64), 32), ((96, 128), 32), ((64, 64), 16), ((80, 48), 8)])
def test_forward_fp32_logits_and_argmax(cuda, size, os_):
    from iseg_amd.data import synthetic_batch

    model = _setup(torch.float32, size, os_)
    x, _ = synthetic_batch(2, size[0], size[1], seed=3)
    with torch.no_grad():
        logits = model(x.cuda(), training=False)[0]
    w = OM.export_weights(model)
    ref = OM.convnext_aspp_forward(w, x.double(), training=False, output_stride=os_)
    err = (logits.cpu().double() - ref["logits"]).abs().max().item()
    assert logits.dtype == torch.float32 and tuple(logits.shape) == (2, size[0], size[1], 21)
    assert err < 1e-3, f"max |logit err| {err}"
    assert torch.equal(logits.argmax(-1).cpu(), O.argmax_first(ref["logits"])), "argmax masks differ"


def test_forward_training_mode_batch_stats_and_moving_update(cuda):
    from iseg_amd.data import synthetic_batch

    model = _setup(torch.float32)
    x, _ = synthetic_batch(4, 64, 64, seed=5)
    w = OM.export_weights(model)
    with torch.no_grad():
        logits = model(x.cuda(), training=True)[0]
    new_stats = {}
    ref = OM.convnext_aspp_forward(w, x.double(), training=True, new_stats=new_stats)
    assert (logits.cpu().double() - ref["logits"]).abs().max().item() < 1e-3
    after = OM.export_weights(model)
    for k, v in new_stats.items():
        assert (after[k] - v).abs().max().item() < 1e-4, k


def test_backward_fp32_matches_oracle_autograd(cuda):
    from iseg_amd import functional as F
    from iseg_amd.data import synthetic_batch

    model = _setup(torch.float32, drop_path=0.1)
    N = 3
    x, y = synthetic_batch(N, 64, 64, seed=7)
    # inject the per-sample drop-path factors (RNG streams cannot match TF's; the arithmetic must)
    g = torch.Generator().manual_seed(11)
    factors = []
    blocks = [b for st in model.backbone.stages for b in st.blocks]
    for b in blocks:
        keep = 1.0 - b.drop_path_prob
        f = torch.floor(keep + torch.rand(N, generator=g)) / keep if b.drop_path_prob > 0 else torch.ones(N)
        b.drop_path_mask = f.float().cuda() if b.drop_path_prob > 0 else None
        factors.append(f.double() if b.drop_path_prob > 0 else None)
    w = OM.export_weights(model)
    model._iseg_store.zero_grad()
    logits = model(x.cuda(), training=True)[0]
    loss = F.softmax_ce_mean(logits, y.cuda(), 21, 255)
    loss.backward()
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items() if not k.endswith(("moving_mean", "moving_variance"))}
    wr.update({k: v for k, v in w.items() if k.endswith(("moving_mean", "moving_variance"))})
    ref = OM.convnext_aspp_forward(wr, x.double(), training=True, dp_factors=factors)
    ref_loss = OM.mean_ce_loss(ref["logits"], y)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    worst = []
    gmax = max(wr[p.iseg_name].grad.abs().max().item() for p in model.parameters())
    for p in model.parameters():
        gr = wr[p.iseg_name].grad
        got = p.grad.detach().cpu().double()
        # some gradients are analytically ~0 (e.g. the bias in front of the BN-normalised ASPP branches: BN's backward sums
        # to zero over the batch), so the floor of the scale is tied to the overall gradient magnitude
        scale = max(gr.abs().max().item(), 1e-3 * gmax)
        worst.append(((got - gr).abs().max().item() / scale, p.iseg_name))
    worst.sort(reverse=True)
    assert worst[0][0] < 5e-3, worst[:5]


@pytest.mark.parametrize("eps,tau", [(1e-4, 0.0), (1e-7, 1e-4)])
def test_train_steps_follow_oracle_adamw(cuda, eps, tau):
    """4 optimisation steps (3 at the well-conditioned epsilon) on a fixed batch reproduce the restatement's loss curve to 1e-3 (fp32, no dropout /
    drop-path), at a well-conditioned epsilon and at Keras' default 1e-7 (optimizers/modern/adamw.py:13-59).  At 1e-7 Adam is scale-free -- an element whose
    true gradient is below the fp32 rounding noise of the sums behind it moves by +-lr with the noise's sign -- so the gradient elements below
    tau = 1e-4 of the model's largest (the oracle's fp64 gradient decides, every step) are zeroed on BOTH sides
    (TrainableModel.gradient_transformers); oracle/models.py ConvNeXtASPPAdamWSteps lists the measured sensitivity to tau."""
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.data import synthetic_batch
    from iseg_amd.distribution.distribution_utils import Strategy
    from iseg_amd.trainer import TrainableModel
    from iseg_amd.utils.train_utils import exclude_no_weight_decay_layers_in_optimizer, get_no_weight_decay_layers_names_from_model
    import re

    model = _setup(torch.float32)
    x, y = synthetic_batch(2, 64, 64, seed=9)
    strat = Strategy(one_device=True)
    opt = get_optimizer(strat, initial_lr=2e-3, end_lr=0.0, epoch_steps=10, train_epoch=1, optimizer="adamw", adamw_weight_decay=0.05)
    exclude_no_weight_decay_layers_in_optimizer(opt, model, print_excluded_list=False)
    opt.epsilon = eps
    tm = TrainableModel(model, optimizer=opt, loss=model.custom_losses(21, 255, 2), loss_weights=model.custom_losses_weights(),
                        metrics=model.custom_metrics(21, 255))
    excl = get_no_weight_decay_layers_names_from_model(model)
    params = {p.iseg_name: p for p in model.parameters()}
    oracle = OM.ConvNeXtASPPAdamWSteps(OM.export_weights(model), x.double(), y, list(params),
                                       lambda s: O.warmup_poly_decay(s, 2e-3, 10, end_lr=0.0, warmup_steps=0, warmup_lr=0.0, power=0.9),
                                       lambda k: 0.0 if any(re.search(n, k) for n in excl) else 0.05, eps=eps, tau=tau)
    masks = {}

    def keep_well_conditioned(store):
        for name, m in masks.items():
            params[name].grad.mul_(m.to(torch.float32).cuda().reshape(params[name].grad.shape))

    tm.gradient_transformers.append(keep_well_conditioned)
    got, want = [], []
    xc, yc = x.cuda(), y.cuda()
    for step in range(4 if eps < 1e-5 else 3):      # (the fp64 oracle step is 11 s of host time; the SGD test below runs the survey's five steps)
        loss, step_masks = oracle.forward_backward()
        want.append(loss)
        masks.clear()
        masks.update(step_masks or {})
        got.append(float(tm.train_step(xc, yc)[0]))
        oracle.apply()
    rel = [abs(a - b) / max(abs(b), 1e-6) for a, b in zip(got, want)]
    assert max(rel) < 1e-3, (got, want)
    assert got[-1] < got[0]


@pytest.mark.parametrize("size,steps", [(64, 5), (512, 3)])      # 512 x 512 = the crop BASELINE configs[1] is quoted on (round 6: three steps there)
def test_train_steps_follow_oracle_sgd(cuda, size, steps):
    """SURVEY section 8(c)'s bar: 5 optimisation steps on a fixed batch (fp32 storage, drop-path / dropout 0, SGD with momentum 0.9 as
    optimizers/modern/sgd.py:12-51, NO gradient mask -- 0 % of the gradient elements are masked) reproduce the restatement's loss curve to 1e-4
    relative and its weight movement to 1e-3.  SGD divides by nothing, so this separates "a gradient is slightly wrong" from "the AdamW comparison
    is ill-conditioned" (test_train_steps_follow_oracle_adamw); the HIP side is bit-reproducible (tests/test_graph_train_gpu.py), so the residual
    is a property of the arithmetic, not of the run.  Learning rate 5e-3: at 2e-2 this model's optimisation is unstable (the fp32-vs-fp64
    rounding difference of step 1 grows 7x per step on BOTH sides of any comparison: measured weight-movement gap 1e-4, 3e-5, 2e-4, ..., 1.2e-2 after
    five steps with the loss still inside 1e-4), which is a property of the problem, not of either implementation."""
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.data import synthetic_batch
    from iseg_amd.distribution.distribution_utils import Strategy
    from iseg_amd.trainer import TrainableModel

    model = _setup(torch.float32, size=(size, size))
    x, y = synthetic_batch(2, size, size, seed=9)
    lr0 = 5e-3
    opt = get_optimizer(Strategy(one_device=True), initial_lr=lr0, end_lr=0.0, epoch_steps=10, train_epoch=1, optimizer="sgd", sgd_momentum_rate=0.9)
    tm = TrainableModel(model, optimizer=opt, loss=model.custom_losses(21, 255, 2), loss_weights=model.custom_losses_weights(),
                        metrics=model.custom_metrics(21, 255))
    params = {p.iseg_name: p for p in model.parameters()}
    w0 = {k: v.detach().cpu().double().clone() for k, v in params.items()}
    oracle = OM.ConvNeXtASPPSGDSteps(OM.export_weights(model), x.double(), y, list(params),
                                     lambda s: O.warmup_poly_decay(s, lr0, 10, end_lr=0.0, warmup_steps=0, warmup_lr=0.0, power=0.9), momentum=0.9)
    got, want = [], []
    xc, yc = x.cuda(), y.cuda()
    for step in range(steps):
        want.append(oracle.forward_backward()[0])
        got.append(float(tm.train_step(xc, yc)[0]))
        if step == 0:      # per-variable gradients of step 1 against fp64, worst first: the message of a failure names the variable
            worst = []
            gmax = max(g.abs().max().item() for g in oracle._pending[0].values())
            for k, p in params.items():
                gr = oracle._pending[0][k]
                d = (p.grad.detach().cpu().double().reshape(gr.shape) - gr).abs().max().item()
                worst.append((d / max(gr.abs().max().item(), 1e-3 * gmax), k))
            worst.sort(reverse=True)
            assert worst[0][0] < 2e-3, worst[:5]
        oracle.apply()
    rel = [abs(a - b) / max(abs(b), 1e-6) for a, b in zip(got, want)]
    assert max(rel) < 1e-5, (rel, got, want)
    assert got[-1] < got[0]
    num = sum(((params[k].detach().cpu().double().reshape(oracle.w[k].shape) - oracle.w[k]) ** 2).sum().item() for k in params)
    den = sum(((oracle.w[k] - w0[k].reshape(oracle.w[k].shape)) ** 2).sum().item() for k in params)
    assert (num / den) ** 0.5 < 1e-3, (num / den) ** 0.5


def test_bf16_forward_close_and_training_decreases_loss(cuda):
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.data import synthetic_batch
    from iseg_amd.distribution.distribution_utils import Strategy
    from iseg_amd.trainer import TrainableModel

    model = _setup(torch.bfloat16, drop_path=0.1)
    x, y = synthetic_batch(2, 64, 64, seed=13)
    with torch.no_grad():
        logits = model(x.cuda(), training=False)[0]
    w = OM.export_weights(model)
    ref = OM.convnext_aspp_forward(w, x.double(), training=False)["logits"]
    scale = ref.abs().max().item()
    err = (logits.cpu().double() - ref).abs().max().item()
    assert err < 0.06 * scale, (err, scale)
    agree = (logits.argmax(-1).cpu() == ref.argmax(-1)).float().mean().item()
    assert agree > 0.97, agree
    opt = get_optimizer(Strategy(one_device=True), initial_lr=1e-3, epoch_steps=100, train_epoch=1, optimizer="adamw")
    tm = TrainableModel(model, optimizer=opt, loss=model.custom_losses(21, 255, 2), loss_weights=model.custom_losses_weights(),
                        metrics=model.custom_metrics(21, 255))
    losses = [float(tm.train_step(x.cuda(), y.cuda())[0]) for _ in range(8)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    res = tm.metric_results()
    assert 0.0 <= res["output_1_IOU"] <= 1.0


def test_loss_kernel_updates_the_running_miou_exactly_like_the_separate_pass(cuda):
    """trainer: when loss and metric agree on classes / ignore label and the labels are at the logits size, the CE kernel also fills the
    MeanIOU confusion matrix (iseg_softmax_ce_confusion); the matrix must equal the one of the separate argmax pass bit for bit"""
    from iseg_amd import kernels as K
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.data import synthetic_batch
    from iseg_amd.distribution.distribution_utils import Strategy
    from iseg_amd.trainer import TrainableModel

    model = _setup(torch.float32, drop_path=0.0)
    x, y = synthetic_batch(2, 64, 64, seed=21)
    opt = get_optimizer(Strategy(one_device=True), initial_lr=1e-3, epoch_steps=100, train_epoch=1, optimizer="adamw")
    losses = model.custom_losses(21, 255, 2)
    tm = TrainableModel(model, optimizer=opt, loss=losses, loss_weights=model.custom_losses_weights(), metrics=model.custom_metrics(21, 255))
    fn = tm._loss_fn(0)
    assert getattr(fn, "confusion_spec", None) == (21, 255)
    with torch.no_grad():
        logits = model(x.cuda(), training=True)[0]      # same weights, same batch statistics as the step below
    want = torch.zeros(21 * 21, dtype=torch.int64, device="cuda")
    K.argmax_confusion(logits.reshape(-1, 21).float().contiguous(), y.cuda().reshape(-1).to(torch.int32), 255, cm=want)
    tm.reset_metrics()
    tm.train_step(x.cuda(), y.cuda())
    got = tm._metrics_for(0)[0].metric.total_cm
    assert int(got.sum()) == int((y != 255).sum())
    assert torch.equal(got.cpu(), want.cpu())


@pytest.mark.parametrize("C,ignore", [(21, 255), (150, 255), (19, 0)])
def test_softmax_ce_confusion_kernel(cuda, C, ignore):
    from iseg_amd import kernels as K

    P = 4 * 33 * 29
    g = torch.Generator().manual_seed(9)
    z = (torch.randn(P, C, generator=g) * 2).float()
    z[::5, 2] = z[::5].max(-1).values      # ties: the first maximal index wins
    y = torch.randint(0, C if ignore != 0 else C + 1, (P,), generator=g, dtype=torch.int32)
    y[torch.rand(P, generator=g) < 0.1] = ignore
    cm = torch.zeros(C * C, dtype=torch.int64, device="cuda")
    px, sm, dz = K.softmax_ce_ignore(z.cuda(), y.cuda(), ignore, want_px=True, want_sum=True, sum_scale=1.0 / P, want_grad=True,
                                     grad_scale=1.0 / P, cm=cm)
    px0, sm0, dz0 = K.softmax_ce_ignore(z.cuda(), y.cuda(), ignore, want_px=True, want_sum=True, sum_scale=1.0 / P, want_grad=True,
                                        grad_scale=1.0 / P)
    assert torch.equal(px, px0) and torch.equal(dz, dz0) and torch.equal(sm, sm0)
    want = torch.zeros(C * C, dtype=torch.int64, device="cuda")
    yy, ig = (y - 1, -1) if ignore == 0 else (y, ignore)      # metrics/seg_metric_wrapper.py:56-59
    K.argmax_confusion(z.cuda(), yy.cuda(), ig, cm=want)
    assert torch.equal(cm.cpu(), want.cpu())
