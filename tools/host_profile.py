#!/usr/bin/env python3
"""cProfile of the host side of one configuration's step (finds Python / launch-path overhead when a step is CPU-bound).
usage: python tools/host_profile.py cfg4_infer [steps]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools import bench_configs as B  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_infer"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from iseg_amd import heads
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_inference import inference_with_sliding_window
    from iseg_amd.data import synthetic_batch
    from iseg_amd.modelhelper import model_common_setup

    common_env_setup(use_one_device_strategy=True, mixed_precision=True, random_seed=0)
    factory, size, batch, training, _ = B.CONFIGS[name]
    model = getattr(heads, factory)(build_input_size=(512, 512))
    model_common_setup(model, restore_checkpoint=False)
    x, _ = synthetic_batch(batch, size, size, seed=7)
    x = x.cuda()

    def step():
        with torch.no_grad():
            return inference_with_sliding_window(x, model, training=False, windows_size=(512, 512))

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
