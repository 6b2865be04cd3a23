#!/usr/bin/env python3
"""Where the data-parallel plumbing costs time on ONE GPU (world of one rank, ISEG_DIST_SINGLE_RANK_COLLECTIVES=1): the flagship step with
all collectives, without the gradient buckets, without the SyncBN messages, and with neither.   python tools/dp_overhead.py [steps]
With a second argument `native`: the stream-ordered exchange (ISEG_DIST_NATIVE=1, collectives through the C ABI's RCCL communicator) --
eager and replayed from one HIP graph, against the same step with every collective stubbed out."""
import os
import sys
import time

os.environ["ISEG_DIST_SINGLE_RANK_COLLECTIVES"] = "1"
NATIVE = len(sys.argv) > 2 and sys.argv[2] == "native"
if NATIVE:
    os.environ["ISEG_DIST_NATIVE"] = "1"
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

    if NATIVE:
        from iseg_amd.graphs import GraphedTrainStep

        real = dist._stream_all_reduce
        count = [0]

        def run(name, step_fn):
            for _ in range(6):
                step_fn(x, y)
            torch.cuda.synchronize()
            count[0] = 0
            t0 = time.perf_counter()
            for _ in range(steps):
                step_fn(x, y)
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            ta = time.perf_counter() - t0
            print(f"{name:52s} {ta / steps * 1e3:6.2f} ms/step  (host enqueue {th / steps * 1e3:5.2f})  collectives enqueued by the host {count[0] / steps:.0f} per step", flush=True)

        def counted(t, raw, *which):
            count[0] += 1
            return real(t, raw, *which)

        dist._stream_all_reduce = counted
        run("stream-ordered exchange, eager", trainer.train_step)
        g = GraphedTrainStep(trainer, warmup=0)
        assert g._eligible(x)
        run("stream-ordered exchange, ONE HIP graph per step", g)
        dist._stream_all_reduce = lambda t, raw, *which: None
        run("collectives stubbed (plumbing only), eager", trainer.train_step)
        g2 = GraphedTrainStep(trainer, warmup=0)
        run("collectives stubbed, ONE HIP graph per step", g2)
        dist._stream_all_reduce = counted
        run("stream-ordered exchange, eager (again)", trainer.train_step)
        return
    variant("all collectives", True, True)
    variant("no gradient buckets", True, False)
    variant("no SyncBN messages", False, True)
    variant("no collectives (plumbing only)", False, False)
    variant("host spin 75 us per bucket instead", False, False, spin_us=75)
    variant("all collectives (again)", True, True)


if __name__ == "__main__":
    main()
