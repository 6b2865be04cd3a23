"""Fused logits tail (csrc/loss.hip iseg_upsample_ce; reference layers/core_model_ext.py:199-256 + losses/catecrossentropy_ignore_label.py:44-88
+ metrics/seg_metric_wrapper.py:89-102): bilinear upsample + ignore-label CE + gradient through the resize + confusion matrix in one
pass, against (a) the fp64 oracle with autograd through its tf.image.resize restatement and (b) the three separate HIP kernels."""
import pytest
import torch

from oracle import tf_ops as O

pytestmark = pytest.mark.gpu

# N, Hi, Wi, C, sy, sx
CASES = [(2, 4, 4, 21, 32, 32), (1, 3, 5, 21, 16, 16), (2, 2, 3, 19, 8, 8), (1, 1, 1, 21, 4, 4), (1, 5, 4, 8, 2, 2), (1, 3, 3, 32, 6, 4),
         (1, 2, 2, 21, 64, 64), (3, 16, 16, 21, 4, 4), (1, 1, 7, 3, 10, 2)]


def _data(N, Hi, Wi, C, sy, sx, seed, ignore=255):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(N, Hi, Wi, C, generator=g) * 2.0
    Ho, Wo = Hi * sy, Wi * sx
    y = torch.randint(0, C, (N, Ho, Wo), generator=g, dtype=torch.int32)
    drop = torch.rand(N, Ho, Wo, generator=g) < 0.1
    y[drop] = ignore
    return z, y, Ho, Wo


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,Hi,Wi,C,sy,sx", CASES)
def test_upsample_ce_matches_oracle_and_separate_kernels(cuda, dtype, N, Hi, Wi, C, sy, sx):
    from iseg_amd import kernels as K

    z, y, Ho, Wo = _data(N, Hi, Wi, C, sy, sx, 7 + Hi + C)
    assert K.upsample_ce_supported(Hi, Wi, Ho, Wo, C)
    zq = z.to(dtype)
    P = N * Ho * Wo
    cm = torch.zeros(C * C, dtype=torch.int64, device="cuda")
    s, dz = K.upsample_ce(zq.cuda(), y.cuda(), Ho, Wo, 255, sum_scale=1.0 / P, grad_scale=1.0 / P, cm=cm)
    # (a) oracle: fp64, autograd through the resize
    zr = zq.double().requires_grad_(True)
    up = O.resize_bilinear(zr, (Ho, Wo))
    loss = O.softmax_ce_ignore(y, up, C, 255).mean()
    loss.backward()
    assert abs(s.item() - loss.item()) <= 2e-5 * max(1.0, abs(loss.item()))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    err = (dz.cpu().double() - zr.grad).abs().max().item()
    assert err <= tol * max(zr.grad.abs().max().item(), 1e-12), (err, zr.grad.abs().max().item())
    # confusion matrix against the fp32 resize of the oracle (ties are measure-zero with random logits)
    cm_ref = O.confusion_matrix(y, O.argmax_first(O.resize_bilinear(zq.float(), (Ho, Wo))), C, 255)
    got = cm.cpu().reshape(C, C).double()
    assert (got - cm_ref).abs().sum().item() <= 2, (got - cm_ref).abs().sum().item()      # an fp32-vs-fp32 near-tie may move a pixel
    assert got.sum().item() == (y != 255).sum().item()
    # (b) the materialised route through the three separate kernels: same loss, gradient and counts
    full = K.resize_bilinear(zq.cuda(), Ho, Wo, out_dtype=torch.float32)
    cm2 = torch.zeros(C * C, dtype=torch.int64, device="cuda")
    _, s2, dl = K.softmax_ce_ignore(full.reshape(-1, C), y.cuda().reshape(-1), 255, want_px=False, want_sum=True, sum_scale=1.0 / P,
                                    want_grad=True, grad_scale=1.0 / P, cm=cm2)
    dz2 = K.resize_bilinear_bwd(dl.reshape(N, Ho, Wo, C), Hi, Wi, dtype)
    assert torch.equal(cm, cm2)
    assert abs(s.item() - s2.item()) <= 1e-5 * max(1.0, abs(s2.item()))
    d = (dz.float() - dz2.float()).abs().max().item()
    assert d <= (1e-5 if dtype == torch.float32 else 1e-2) * max(dz2.float().abs().max().item(), 1e-12)


@pytest.mark.parametrize("ignore", [255, 0])
def test_upsample_ce_class_weights_ignore_zero_and_determinism(cuda, ignore):
    from iseg_amd import kernels as K

    N, Hi, Wi, C, sy, sx = 2, 3, 4, 21, 16, 16
    z, y, Ho, Wo = _data(N, Hi, Wi, C, sy, sx, 3, ignore=ignore)
    if ignore == 0:
        y = torch.randint(0, C + 1, y.shape, generator=torch.Generator().manual_seed(1), dtype=torch.int32)      # 0 = ignored, 1..C -> classes
    cw = torch.rand(C, generator=torch.Generator().manual_seed(2)) + 0.5
    P = N * Ho * Wo
    outs = []
    for _ in range(2):
        s, dz = K.upsample_ce(z.cuda(), y.cuda(), Ho, Wo, ignore, class_w=cw.cuda(), sum_scale=1.0 / P, grad_scale=1.0 / P)
        outs.append((s.clone(), dz.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    zr = z.double().requires_grad_(True)
    loss = O.softmax_ce_ignore(y, O.resize_bilinear(zr, (Ho, Wo)), C, ignore, cw.double()).mean()
    loss.backward()
    assert abs(outs[0][0].item() - loss.item()) <= 2e-5 * max(1.0, abs(loss.item()))
    assert (outs[0][1].cpu().double() - zr.grad).abs().max().item() <= 2e-5 * zr.grad.abs().max().item()


def test_upsample_ce_refuses_other_geometry(cuda):
    from iseg_amd import _hip, kernels as K

    assert not K.upsample_ce_supported(4, 4, 12, 12, 21)        # x3: odd factor
    assert not K.upsample_ce_supported(4, 4, 130, 128, 21)      # not an integer factor
    assert not K.upsample_ce_supported(4, 4, 128, 128, 150)     # too many classes for the register-resident form
    assert not K.upsample_ce_supported(2, 2, 256, 256, 21)      # x128 columns
    z = torch.zeros(1, 4, 4, 21, device="cuda")
    y = torch.zeros(1, 12, 12, dtype=torch.int32, device="cuda")
    with pytest.raises(_hip.HipCallError):
        K.upsample_ce(z, y, 12, 12, 255)


def test_trainer_uses_the_fused_tail_and_matches_the_materialised_route(cuda):
    """two identical models, one step each: CoreTrain's step with the deferred upsample vs the same step with the full-resolution
    logits written out -- same loss, same confusion matrix, same parameter gradients (fp32 storage)"""
    import iseg_amd.functional as F
    from iseg_amd import nn
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.core_train import CoreTrain
    from iseg_amd.data import synthetic_batch
    from iseg_amd.heads import convnext_tiny_aspp
    from iseg_amd.modelhelper import model_common_setup

    strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=False, random_seed=0)
    x, y = synthetic_batch(2, 64, 64, seed=0)
    x, y = x.cuda(), y.cuda()
    results = []
    for fused in (True, False):
        nn.set_seed(0)
        model = convnext_tiny_aspp(num_class=21, build_input_size=(64, 64), drop_path_rate=0.0, dropout_rate=0.0, layer_scale_init_value=1.0)
        helper = model_common_setup(model, restore_checkpoint=False)
        helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-3, optimizer="adamw", epoch_steps=10, train_epoch=1))
        trainer = CoreTrain(helper, None).create_trainable_model(21, batch_size=2)
        calls = []
        real = F.upsample_softmax_ce_mean
        F.upsample_softmax_ce_mean = (lambda *a, **k: (calls.append(1), real(*a, **k))[1]) if fused else None
        try:
            if not fused:      # force the materialised route
                saved = F.DeferredLogits.fusable
                F.DeferredLogits.fusable = lambda self, nc: False
            loss = float(trainer.train_step(x, y)[0].detach())
        finally:
            F.upsample_softmax_ce_mean = real
            if not fused:
                F.DeferredLogits.fusable = saved
        assert bool(calls) == fused
        cmv = None
        for m in trainer._metrics_for(0):
            cmv = m.metric.total_cm.clone()
        results.append((loss, cmv, model._iseg_store.flat_g.clone()))
    (l0, c0, g0), (l1, c1, g1) = results
    assert abs(l0 - l1) <= 1e-5 * abs(l1)
    assert c0 is not None and torch.equal(c0, c1)
    assert (g0 - g1).abs().max().item() <= 1e-4 * g1.abs().max().item()      # the gradients the optimizer consumed
