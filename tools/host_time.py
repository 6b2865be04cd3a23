#!/usr/bin/env python3
"""Host-side enqueue time of the flagship train step next to its GPU time: how much head-room the Python host has before the step turns
host-bound (ISEG_DIST_SINGLE_RANK_COLLECTIVES=1 adds the data-parallel plumbing on one GPU).   python tools/host_time.py [steps] [cfg1|cfg3|cfg4|cfg5|v2]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def build_config(name):
    """one of tools/bench_configs.py's training configurations instead of the flagship"""
    from iseg_amd import heads
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.core_train import CoreTrain
    from iseg_amd.data import synthetic_batch
    from iseg_amd.modelhelper import model_common_setup
    from tools.bench_configs import CONFIGS

    factory, size, batch, training, _ = CONFIGS[name]
    strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=True, random_seed=0)
    model = getattr(heads, factory)(build_input_size=(size, size))
    helper = model_common_setup(model, restore_checkpoint=False)
    helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-4, end_lr=0.0, epoch_steps=1000, train_epoch=30, optimizer="adamw",
                                       adamw_weight_decay=0.05))
    trainer = CoreTrain(helper, None).create_trainable_model(21, ignore_label=255, batch_size=batch)
    x, y = synthetic_batch(batch, size, size, seed=7)
    return trainer, x.cuda(), y.cuda()


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    config = sys.argv[2] if len(sys.argv) > 2 else None
    sys.argv = [sys.argv[0]]
    if config:
        trainer, x, y = build_config(config)
    else:
        args = bench.parse()
        from iseg_amd.data import synthetic_batch

        strategy, model, trainer = bench.build_trainer(args)
        x, y = synthetic_batch(args.batch, args.size, args.size, seed=100)
        x, y = x.cuda(), y.cuda()
    for _ in range(8):
        trainer.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_step(x, y)
    t_host = time.perf_counter() - t0          # the loop returns when everything is enqueued
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"host enqueue {t_host / steps * 1e3:.2f} ms/step, with the GPU drained {t_all / steps * 1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
