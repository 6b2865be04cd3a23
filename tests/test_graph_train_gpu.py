"""Self-comparison properties of the training step (they run LAST, tests/conftest.py): the step is bit-reproducible -- the reference's default is
`use_deterministic=True` -> `enable_op_determinism()` (core_env.py:39-48) and no kernel of the step uses a floating-point atomic any more (round 4:
every cross-lane / cross-workgroup sum runs in a fixed order) -- so two eager runs from the same state give the SAME BITS, and the HIP-graph
replay of the step (iseg_amd/graphs.py GraphedTrainStep) gives the same bits as the eager step: same drop-path / dropout draws (device-resident
draw counter), same learning-rate schedule (device-resident optimizer scalars), same running confusion matrix.  No tolerance in this file."""
import pytest
import torch

pytestmark = pytest.mark.gpu


OPTIMIZER = ["adamw"]
CLIPNORM = [None]


FACTORY = ["convnext_tiny_aspp"]


def _trainer(seed=3):
    from iseg_amd import heads, nn
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.core_train import CoreTrain
    from iseg_amd.modelhelper import model_common_setup

    strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=True, random_seed=seed)
    if FACTORY[0] == "convnext_tiny_aspp":
        model = heads.convnext_tiny_aspp(build_input_size=(64, 64), drop_path_rate=0.2, dropout_rate=0.1)
    else:
        model = getattr(heads, FACTORY[0])(build_input_size=(64, 64))
    helper = model_common_setup(model, restore_checkpoint=False)
    helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-3 if OPTIMIZER[0] == "adamw" else 2e-2, end_lr=0.0, epoch_steps=20, train_epoch=1,
                                       warmup_steps=3, warmup_lr=1e-5, optimizer=OPTIMIZER[0], adamw_weight_decay=0.05, clipnorm=CLIPNORM[0]))
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


def _batches():
    from iseg_amd.data import synthetic_batch

    out = []
    for s in (5, 6, 7):
        x, y = synthetic_batch(4, 64, 64, seed=s)
        out.append((x.cuda(), y.cuda()))
    return out


@pytest.mark.parametrize("optimizer", ["adamw"])      # (SGD's eager run is compared bit for bit with its replay below)
def test_two_eager_runs_are_bit_identical(cuda, optimizer):
    """nine flagship steps (drop-path 0.2, dropout 0.1, bf16 storage, running mIoU) twice from the same state: identical losses, weights,
    optimizer state and confusion matrix, bit for bit"""
    OPTIMIZER[0] = optimizer
    batches = _batches()
    la, wa, ita, cma, _, w0a = _run(False, 9, batches)
    lb, wb, itb, cmb, _, w0b = _run(False, 9, batches)
    assert torch.equal(w0a, w0b), "the two trainers did not start from the same weights"
    assert ita == itb == 9
    assert la == lb, (la, lb)
    assert torch.equal(wa, wb), float((wa - wb).abs().max())
    assert torch.equal(cma, cmb)


@pytest.mark.parametrize("optimizer", ["adamw", "sgd"])
def test_graphed_train_steps_follow_eager(cuda, optimizer):
    """replayed steps == eager steps, bit for bit: losses of every step, the weights after nine steps, the confusion matrix, the schedule"""
    OPTIMIZER[0] = optimizer
    batches = _batches()
    le, we, ite, cme, _, w0e = _run(False, 9, batches)
    lg, wg, itg, cmg, step, w0g = _run(True, 9, batches)
    assert torch.equal(w0e, w0g), "the two trainers did not start from the same weights"
    assert any(e.get("graph") is not None for e in step.entries.values()), "the step was never captured"
    assert ite == itg == 9
    assert le == lg, (le, lg)
    assert len(set(lg)) == len(lg)      # drop-path / dropout draws and the batches differ from step to step
    assert torch.equal(we, wg), float((we - wg).abs().max())
    assert cme is not None and torch.equal(cme, cmg)
    # the optimizer scalars the last replay read: learning rate of step 9 of the warm-up + poly schedule
    opt = step.tm.optimizer
    opt.iterations -= 1
    want_lr = opt.current_lr()
    opt.iterations += 1
    assert abs(float(opt._hp_fixed[0]) - want_lr) <= 1e-7 * want_lr


def test_batches_written_into_the_captured_input_buffers_replay_the_same_steps(cuda):
    """round 6: GraphedTrainStep.input_buffers() hands out the captured step's own input tensors; a producer that writes the next batch there and
    passes THEM saves the per-step device-to-device copy (36 us of the flagship step) -- the steps are bit for bit the copying ones"""
    from iseg_amd import functional as F
    from iseg_amd.graphs import GraphedTrainStep

    OPTIMIZER[0] = "adamw"
    batches = _batches()
    lc, wc, _, cmc, _, w0c = _run(True, 8, batches)
    F._RNG_COUNTER[0] = 0
    F._DROP_PATH_POOL.__init__()
    tm = _trainer()
    assert torch.equal(w0c, tm.store.flat_w)
    step = GraphedTrainStep(tm, warmup=2)
    losses = []
    for i in range(8):
        x, y = batches[i % len(batches)]
        bufs = step.input_buffers(x, y)
        if bufs is None:
            assert i <= 2      # two warm-up calls + the capturing call
        else:
            bufs[0].copy_(x)
            bufs[1].copy_(y)
            x, y = bufs
        losses.append(float(step(x, y)[0]))
    torch.cuda.synchronize()
    assert losses == lc, (losses, lc)
    assert torch.equal(tm.store.flat_w, wc)


def test_two_input_signatures_share_the_optimizer_slot(cuda):
    """a second input signature (here: a smaller last batch) captures a second graph; both graphs read the optimizer's scalars -- learning rate,
    bias correction -- from the SAME fixed device slot, so replays of the first graph after the second capture still follow the eager run bit for
    bit (a fresh slot per capture left the first graph reading a freed address: round-3 advisor finding)"""
    from iseg_amd.data import synthetic_batch

    OPTIMIZER[0] = "adamw"
    big, small = [], []
    for s in (5, 6):
        x, y = synthetic_batch(4, 64, 64, seed=s)
        big.append((x.cuda(), y.cuda()))
        x, y = synthetic_batch(2, 64, 64, seed=10 + s)
        small.append((x.cuda(), y.cuda()))
    order = [big[0], big[1], big[0], small[0], small[1], small[0], big[1], small[1], big[0], big[1], small[0]]
    le, we, ite, cme, _, w0e = _run(False, len(order), order)
    lg, wg, itg, cmg, step, w0g = _run(True, len(order), order)
    assert torch.equal(w0e, w0g)
    graphs = [e for e in step.entries.values() if e.get("graph") is not None]
    assert len(graphs) == 2, "both signatures must have been captured"
    assert le == lg, (le, lg)
    assert torch.equal(we, wg), float((we - wg).abs().max())
    assert torch.equal(cme, cmg)


def test_graphed_step_with_per_variable_norm_clipping(cuda):
    """clipnorm (Keras' per-variable norm clip, AdamW_EXT._clip_gradients with its NaN scrub: optimizers/modern/adamw.py:63-74) adds the squared-norm
    reduction in front of the step kernel; captured with the rest of the step it must replay the eager run bit for bit, and it must bite (the curve
    differs from the unclipped one)"""
    OPTIMIZER[0] = "adamw"
    batches = _batches()
    try:
        CLIPNORM[0] = None
        plain = _run(False, 6, batches)[0]
        CLIPNORM[0] = 0.05
        le, we, _, cme, _, w0e = _run(False, 6, batches)
        lg, wg, _, cmg, step, w0g = _run(True, 6, batches)
    finally:
        CLIPNORM[0] = None
    assert any(e.get("graph") is not None for e in step.entries.values()), "the step was never captured"
    assert le == lg, (le, lg)
    assert torch.equal(we, wg) and torch.equal(cme, cmg)
    assert le != plain, "the clip changed nothing: too loose to exercise the norm reduction"


@pytest.mark.parametrize("factory", ["swin_tiny_fpn", "intern_image_base_aspp"])
def test_other_compositions_replay_bit_exact(cuda, factory):
    """Swin-T + FPN (row-table LayerNorm paired with the residual gather through a link object, the MLP halves on the fused ConvNeXt-MLP node,
    FPN levels as one node with training-mode BatchNorm) and InternImage-B + ASPP (post-norm tails, DCNv3 fixed-point windows): four replayed
    steps == four eager steps, bit for bit -- the Python-side pairing of the new nodes exists only while the step is captured"""
    FACTORY[0] = factory
    try:
        batches = _batches()
        le, we, ite, cme, _, w0e = _run(False, 4, batches)
        lg, wg, itg, cmg, step, w0g = _run(True, 4, batches)
    finally:
        FACTORY[0] = "convnext_tiny_aspp"
    assert torch.equal(w0e, w0g), "the two trainers did not start from the same weights"
    assert any(e.get("graph") is not None for e in step.entries.values()), "the step was never captured"
    assert le == lg, (le, lg)
    assert torch.equal(we, wg), float((we - wg).abs().max())
    assert cme is not None and torch.equal(cme, cmg)
