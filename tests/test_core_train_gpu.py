"""CoreTrain.train() end to end (core_train.py:74-167 of the reference): dataset -> map(inputs_process) -> shuffle -> repeat -> batch ->
prefetch -> fit -> callbacks (CheckpointSaver per epoch, ModelCallback, TimeCallback) -> validation -> save_checkpoint, then a FRESH
model restores the newest checkpoint (modelhelper.py:113-264) and must produce identical logits."""
import glob
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(size, seed_shift=0):
    from iseg_amd import nn
    from iseg_amd.heads import convnext_tiny_aspp

    nn.set_seed(seed_shift)
    return convnext_tiny_aspp(num_class=21, build_input_size=size, drop_path_rate=0.0, dropout_rate=0.1, layer_scale_init_value=1.0)


def test_core_train_fit_checkpoint_restore_roundtrip(cuda, tmp_path):
    from iseg_amd import nn
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.core_train import CoreTrain
    from iseg_amd.data import synthetic_batch, synthetic_dataset
    from iseg_amd.modelhelper import model_common_setup

    size = (64, 64)
    ckpt_dir = str(tmp_path / "ckpt")
    strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=False, random_seed=0)
    model = _model(size)
    epochs_seen = []
    model.on_epoch_end = lambda epoch, logs: epochs_seen.append((epoch, dict(logs)))       # SegBase hook driven by ModelCallback
    helper = model_common_setup(model, restore_checkpoint=True, checkpoint_dir=ckpt_dir, max_checkpoints_to_keep=2)
    helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-3, end_lr=0.0, epoch_steps=3, train_epoch=3, optimizer="adamw",
                                       adamw_weight_decay=0.01))
    train_ds = synthetic_dataset(10, size[0], size[1], seed=5)
    val_ds = synthetic_dataset(4, size[0], size[1], seed=9)
    w_before = {p.iseg_name: p.detach().clone() for p in model.parameters()}
    trainer = CoreTrain(helper, train_ds, val_ds, val_image_count=4)
    history = trainer.train(strategy, num_class=21, ignore_label=255, batch_size=2, eval_batch_size=2, shuffle_rate=4, epoch_steps=3,
                            train_epoches=3, verbose=0, validation_freq=1)
    torch.cuda.synchronize()
    # three epochs of three steps ran, every epoch logged a finite loss, train / validation mIoU in [0, 1]
    assert len(history) == 3 and [e for e, _ in epochs_seen] == [0, 1, 2]
    for logs in history:
        assert logs["loss"] == logs["loss"] and 0.0 <= logs["output_1_IOU"] <= 1.0 and 0.0 <= logs["val_output_1_IOU"] <= 1.0
    assert helper.optimizer.iterations == 9
    moved = max(float((p.detach() - w_before[p.iseg_name]).abs().max()) for p in model.parameters())
    assert moved > 1e-5, "the optimizer never changed the weights"
    # CheckpointSaver wrote one file per epoch, rotation kept the newest two
    files = sorted(glob.glob(os.path.join(ckpt_dir, "id-*.ckpt.weights.pt")))
    assert len(files) == 2, files
    x, _ = synthetic_batch(2, size[0], size[1], seed=77)
    with torch.no_grad():
        want = model(x.cuda(), training=False)[0].clone()
    # a fresh model (different init) + restore_checkpoint=True -> the same function, bit for bit (weights AND BN moving statistics)
    other = _model(size, seed_shift=123)
    with torch.no_grad():
        differs = (other(x.cuda(), training=False)[0] - want).abs().max().item()
    assert differs > 1e-3
    helper2 = model_common_setup(other, restore_checkpoint=True, checkpoint_dir=ckpt_dir, max_checkpoints_to_keep=2)
    assert helper2.list_checkpoints()[-1] == files[-1]
    with torch.no_grad():
        got = other(x.cuda(), training=False)[0]
    assert torch.equal(got, want)
    # resume bookkeeping: initial_epoch=-1 is derived from the optimizer's iteration counter (core_train.py:101-102)
    nn.set_seed(0)


def test_restore_refuses_a_mismatching_checkpoint(cuda, tmp_path):
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.modelhelper import ModelHelper

    common_env_setup(use_one_device_strategy=True, mixed_precision=False, random_seed=0)
    a = _model((64, 64))
    ha = ModelHelper(a, str(tmp_path), 3)
    p1 = ha.save_checkpoint()
    p2 = ha.save_checkpoint()          # same second: the counter keeps the names apart
    assert p1 != p2 and len(ha.list_checkpoints()) == 2
    from iseg_amd.heads import resnet50_aspp

    b = resnet50_aspp(num_class=21, build_input_size=(64, 64))
    hb = ModelHelper(b, str(tmp_path), 3)
    with pytest.raises(ValueError):
        hb.restore_checkpoint()
    assert hb.restore_checkpoint(skip_mismatch=True) == p2


def test_jit_compile_replays_the_step_from_a_hip_graph_with_identical_results(cuda, tmp_path):
    """CoreTrain.train(jit_compile=True) (core_train.py:86-91: Keras compiles the train function): here the step is replayed from one HIP graph.
    Same seeds, same data order -> the history (losses, mIoU) and the weights of the compiled run equal the eager run's, bit for bit."""
    from iseg_amd import functional as F
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.core_train import CoreTrain
    from iseg_amd.data import synthetic_dataset
    from iseg_amd.modelhelper import model_common_setup

    size = (64, 64)

    def run(jit):
        F._RNG_COUNTER[0] = 0
        F._DROP_PATH_POOL.__init__()
        strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=True, random_seed=0)
        model = _model(size)
        helper = model_common_setup(model, restore_checkpoint=False, checkpoint_dir=None)
        helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-3, end_lr=0.0, epoch_steps=4, train_epoch=2, optimizer="adamw",
                                           adamw_weight_decay=0.01))
        trainer = CoreTrain(helper, synthetic_dataset(16, size[0], size[1], seed=5), None)
        history = trainer.train(strategy, num_class=21, ignore_label=255, batch_size=2, shuffle_rate=4, epoch_steps=4, train_epoches=2, verbose=0,
                                jit_compile=jit)
        torch.cuda.synchronize()
        return history, {p.iseg_name: p.detach().clone() for p in model.parameters()}

    from iseg_amd import graphs

    captures = []
    real_capture = graphs.GraphedTrainStep._capture
    graphs.GraphedTrainStep._capture = lambda self, x, y: (captures.append(tuple(x.shape)), real_capture(self, x, y))[1]
    try:
        h0, w0 = run(None)
        assert not captures
        h1, w1 = run(True)
    finally:
        graphs.GraphedTrainStep._capture = real_capture
        from iseg_amd import nn

        nn.set_compute_dtype(torch.float32)
    assert len(captures) == 1, f"jit_compile=True must capture the step once, captured {captures}"
    assert h0 == h1, (h0, h1)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k
