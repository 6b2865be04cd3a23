#!/usr/bin/env python3
"""Who launches the runtime's copy / ATen kernels inside one training step of a configuration: torch.profiler with stacks, grouped by the innermost
frame of this package.   python tools/find_copies_step.py cfg3"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from tools.bench_configs import CONFIGS  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    from iseg_amd import heads
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.core_train import CoreTrain
    from iseg_amd.data import synthetic_batch
    from iseg_amd.modelhelper import model_common_setup

    strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=True, random_seed=0)
    factory, size, batch, training, desc = CONFIGS[name]
    model = getattr(heads, factory)(build_input_size=(size, size))
    helper = model_common_setup(model, restore_checkpoint=False)
    x, y = synthetic_batch(batch, size, size, seed=7)
    x, y = x.cuda(), y.cuda()
    helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-4, end_lr=0.0, epoch_steps=1000, train_epoch=30, optimizer="adamw", adamw_weight_decay=0.05))
    trainer = CoreTrain(helper, None).create_trainable_model(21, ignore_label=255, batch_size=batch)
    for _ in range(3):
        trainer.train_step(x, y)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        trainer.train_step(x, y)
        torch.cuda.synchronize()
    allops = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::") or "emcpy" in ev.name or "emset" in ev.name)
    print("every aten / memcpy event of the step:", dict(allops.most_common(40)))
    counts = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::fill_", "aten::zero_", "aten::add_", "aten::mul", "aten::to", "aten::_to_copy",
                       "aten::index", "aten::index_put_", "aten::slice_scatter", "aten::select_scatter"):
            frame = next((f for f in (ev.stack or []) if "iseg_amd" in f), (ev.stack or ["?"])[0] if ev.stack else "?")
            shapes = str(ev.input_shapes)[:60]
            counts[(ev.name, frame.strip()[-110:], shapes)] += 1
    for (n, f, s), c in counts.most_common(40):
        print(f"{c:4d} x {n:18s} {s:62s} {f}")


if __name__ == "__main__":
    main()
