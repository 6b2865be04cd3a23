#!/usr/bin/env python3
"""Host-side enqueue time of the flagship train step next to its GPU time: how much head-room the Python host has before the step turns
host-bound (ISEG_DIST_SINGLE_RANK_COLLECTIVES=1 adds the data-parallel plumbing on one GPU).   python tools/host_time.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    sys.argv = [sys.argv[0]]
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
