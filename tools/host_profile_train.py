#!/usr/bin/env python3
"""cProfile of the host side of a TRAINING step of one tools/bench_configs.py configuration (where does the Python time of a host-bound
step go).   python tools/host_profile_train.py cfg1 [steps]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.host_time import build_config  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    trainer, x, y = build_config(name)
    for _ in range(4):
        trainer.train_step(x, y)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        trainer.train_step(x, y)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
