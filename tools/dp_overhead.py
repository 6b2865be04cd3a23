#!/usr/bin/env python3
"""Where the data-parallel plumbing costs time on ONE GPU (world of one rank, ISEG_DIST_SINGLE_RANK_COLLECTIVES=1): the flagship step with
all collectives, without the gradient buckets, without the SyncBN messages, and with neither.   python tools/dp_overhead.py [steps]"""
import os
import sys
import time

os.environ["ISEG_DIST_SINGLE_RANK_COLLECTIVES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    from iseg_amd import dist
    from iseg_amd.data import synthetic_batch

    strategy, model, trainer = bench.build_trainer(args)
    x, y = synthetic_batch(args.batch, args.size, args.size, seed=100)
    x, y = x.cuda(), y.cuda()
    real_ar = dist.all_reduce_sum
    print("buckets (MiB):", [round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi, _ in trainer.reducer.buckets], flush=True)
    calls = [0, 0]

    def counting(t, async_op=False):
        calls[1 if async_op else 0] += 1
        return real_ar(t, async_op=async_op)

    def variant(name, sync_on, async_on, spin_us=0):
        def ar(t, async_op=False):
            if async_op and spin_us:      # a pure host delay where the bucket would go out: is the backward pass host-bound there?
                t1 = time.perf_counter() + spin_us * 1e-6
                while time.perf_counter() < t1:
                    pass
                return None
            if (async_op and not async_on) or (not async_op and not sync_on):
                return None
            return counting(t, async_op)

        dist.all_reduce_sum = ar
        for _ in range(6):
            trainer.train_step(x, y)
        torch.cuda.synchronize()
        calls[0] = calls[1] = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            trainer.train_step(x, y)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        ta = time.perf_counter() - t0
        print(f"{name:34s} {ta / steps * 1e3:6.2f} ms/step  (host enqueue {th / steps * 1e3:5.2f})  blocking {calls[0] / steps:.0f}  async {calls[1] / steps:.0f} per step",
              flush=True)

    variant("all collectives", True, True)
    variant("no gradient buckets", True, False)
    variant("no SyncBN messages", False, True)
    variant("no collectives (plumbing only)", False, False)
    variant("host spin 75 us per bucket instead", False, False, spin_us=75)
    variant("all collectives (again)", True, True)


if __name__ == "__main__":
    main()
