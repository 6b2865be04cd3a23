"""HIP-graph replay of the training step (iseg_amd/graphs.py GraphedTrainStep): replayed steps follow eager steps from the same state --
same drop-path / dropout draws (device-resident draw counter, checked bit-exactly at the kernels), same learning-rate schedule
(device-resident optimizer scalars), same running confusion matrix.  Two EAGER runs of this step already differ in the last bits (float
LDS atomics in the LayerNorm / column-sum parameter gradients) and Adam turns that into ~1e-4 of the loss within a few
steps, so the step-by-step comparison carries that tolerance; a wrong mask or learning rate moves the loss by 1e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu


OPTIMIZER = ["adamw"]


def _trainer(seed=3):
    from iseg_amd import nn
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.core_train import CoreTrain
    from iseg_amd.heads import convnext_tiny_aspp
    from iseg_amd.modelhelper import model_common_setup

    strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=True, random_seed=seed)
    model = convnext_tiny_aspp(build_input_size=(64, 64), drop_path_rate=0.2, dropout_rate=0.1)
    helper = model_common_setup(model, restore_checkpoint=False)
    helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-3 if OPTIMIZER[0] == "adamw" else 2e-2, end_lr=0.0, epoch_steps=20, train_epoch=1,
                                       warmup_steps=3, warmup_lr=1e-5, optimizer=OPTIMIZER[0], adamw_weight_decay=0.05))
    return CoreTrain(helper, None).create_trainable_model(21, ignore_label=255, batch_size=4)


def _run(graphed, steps, batches):
    from iseg_amd import functional as F
    from iseg_amd.graphs import GraphedTrainStep

    F._RNG_COUNTER[0] = 0
    F._DROP_PATH_POOL.__init__()      # the pool replays the (samples, keep) plan it recorded: both runs must start without one
    tm = _trainer()
    w0 = tm.store.flat_w.clone()
    step = GraphedTrainStep(tm, warmup=2) if graphed else tm.train_step
    losses = []
    for i in range(steps):
        x, y = batches[i % len(batches)]
        out = step(x, y)
        losses.append(float(out[0]))      # (a host sync per step: the captured output buffer is read before the next replay overwrites it)
    torch.cuda.synchronize()
    cm = None
    for ms in tm.metrics.values() if isinstance(tm.metrics, dict) else [tm.metrics]:
        for m in ms:
            cm = m.metric.total_cm.clone()
    return losses, tm.store.flat_w.clone(), tm.optimizer.iterations, cm, step, w0


def test_seed_offset_equals_shifted_seed(cuda):
    """the device-resident addend of the dropout / drop-path seeds: kernel(seed, offset) == kernel(seed + offset), bit for bit"""
    from iseg_amd import kernels as K

    x = torch.randn(4, 33, 17, 64, device="cuda").to(torch.bfloat16)
    keeps = torch.tensor([0.9, 0.8, 0.7], device="cuda")
    seed, off = 0x1234567890ABCDEF, 0xD1B54A32D192ED03 * 7 & 0xFFFFFFFFFFFFFFFF
    shifted = (seed + off) & 0xFFFFFFFFFFFFFFFF
    want = (K.dropout(x, 0.3, shifted), K.drop_path_mask(16, 0.8, shifted, x.device), K.drop_path_masks(keeps, 16, shifted))
    off_t = torch.tensor([off - (1 << 64) if off >= (1 << 63) else off], dtype=torch.int64, device="cuda")
    K.set_seed_offset(off_t)
    try:
        got = (K.dropout(x, 0.3, seed), K.drop_path_mask(16, 0.8, seed, x.device), K.drop_path_masks(keeps, 16, seed))
    finally:
        K.set_seed_offset(None)
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert not torch.equal(got[0], K.dropout(x, 0.3, seed))


@pytest.mark.parametrize("optimizer,loss_tol,weight_tol", [("adamw", 2e-3, 0.5), ("sgd", 5e-4, 0.08)])
def test_graphed_train_steps_follow_eager(cuda, optimizer, loss_tol, weight_tol):
    """AdamW: the flagship's optimizer (its sign-like early steps amplify the last-bit run-to-run differences of the float-atomic reductions,
    so the weights of two EAGER runs already sit ~0.15 of their movement apart: only the loss curve and the schedule are tight there);
    SGD with momentum: no such amplification, the weights must agree closely too -- closely = the eager run-to-run band: the remaining LDS
    float atomics (BatchNorm / depthwise / column-sum parameter gradients) leave two eager runs 0.03-0.05 of their weight movement apart after
    nine steps about one time in three, and bit-identical the other times"""
    from iseg_amd.data import synthetic_batch

    OPTIMIZER[0] = optimizer

    batches = []
    for s in (5, 6, 7):
        x, y = synthetic_batch(4, 64, 64, seed=s)
        batches.append((x.cuda(), y.cuda()))
    le, we, ite, cme, _, w0e = _run(False, 9, batches)
    lg, wg, itg, cmg, step, w0g = _run(True, 9, batches)
    assert torch.equal(w0e, w0g), "the two trainers did not start from the same weights"
    assert any(e.get("graph") is not None for e in step.entries.values()), "the step was never captured"
    assert ite == itg == 9
    for i, (a, b) in enumerate(zip(le, lg)):
        assert abs(a - b) <= loss_tol * abs(a), (i, le, lg)
    assert len(set(lg)) == len(lg)      # drop-path / dropout draws and the batches differ from step to step
    rel = ((we - wg).norm() / (we - w0e).norm()).item()      # the weights moved the same way
    assert rel < weight_tol, rel
    assert cme is not None and int(cme.sum()) == int(cmg.sum())      # every step's pixels were counted once
    # the optimizer scalars the last replay read: learning rate of step 9 of the warm-up + poly schedule
    opt = step.tm.optimizer
    opt.iterations -= 1
    want_lr = opt.current_lr()
    opt.iterations += 1
    assert abs(float(opt._hp_fixed[0]) - want_lr) <= 1e-7 * want_lr
