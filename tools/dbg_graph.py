import sys, os
sys.path.insert(0, "/root/repo")
import torch
from tests.test_graph_train_gpu import _run
import tests.test_graph_train_gpu as T
from iseg_amd.data import synthetic_batch

def trainer_factory(dp, do, sched):
    def _trainer(seed=3):
        from iseg_amd.core_env import common_env_setup
        from iseg_amd.core_optimizer import get_optimizer
        from iseg_amd.core_train import CoreTrain
        from iseg_amd.heads import convnext_tiny_aspp
        from iseg_amd.modelhelper import model_common_setup
        strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=True, random_seed=seed)
        model = convnext_tiny_aspp(build_input_size=(64, 64), drop_path_rate=dp, dropout_rate=do)
        helper = model_common_setup(model, restore_checkpoint=False)
        if sched:
            opt = get_optimizer(strategy, initial_lr=1e-3, end_lr=0.0, epoch_steps=20, train_epoch=1, warmup_steps=3, warmup_lr=1e-5, optimizer="adamw", adamw_weight_decay=0.05)
        else:
            opt = get_optimizer(strategy, initial_lr=1e-3, end_lr=1e-3, epoch_steps=20, train_epoch=1, warmup_steps=0, optimizer="sgd")
        helper.set_optimizer(opt)
        return CoreTrain(helper, None).create_trainable_model(21, ignore_label=255, batch_size=4)
    return _trainer

batches = []
for s in (5, 6, 7):
    x, y = synthetic_batch(4, 64, 64, seed=s)
    batches.append((x.cuda(), y.cuda()))
import io, contextlib
for name, (dp, do, sched, same_batch) in {"full": (0.2, 0.1, True, False), "no_rng": (0.0, 0.0, True, False), "no_rng_sgd_const": (0.0, 0.0, False, False), "rng_sgd_const": (0.2, 0.1, False, False), "no_rng_same_batch": (0.0, 0.0, True, True)}.items():
    T._trainer = trainer_factory(dp, do, sched)
    bs = batches[:1] if same_batch else batches
    with contextlib.redirect_stdout(io.StringIO()):
        le = T._run(False, 8, bs)[0]
        le2 = T._run(False, 8, bs)[0]
        lg = T._run(True, 8, bs)[0]
    diff = [i for i, (a, b) in enumerate(zip(le, lg)) if a != b]
    diff2 = [i for i, (a, b) in enumerate(zip(le, le2)) if a != b]
    print(name, "first mismatch eager/graph:", diff[:3], "eager/eager:", diff2[:3], [f"{a:.6f}/{b:.6f}" for a, b in zip(le, lg)][3:6])
