"""Thread budget of the CPU oracle.  TEST INFRASTRUCTURE (like everything under oracle/): used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.

torch sizes its intra-op pool by the machine's core count, but a container is often capped by a cgroup CPU quota far below it -- the GPU boxes of
this pool show 256 logical cores and a quota of 16 -- and 128 threads time-sliced onto 16 cores run the fp64 oracle 8 x slower than 16 threads do
(tools/threads_probe.py: 2.28 s vs 0.28 s for one depthwise + LayerNorm + pointwise + GELU block, forward and backward)."""
import os


def cpu_budget():
    """cores this process may really use: the affinity mask, capped by the cgroup CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def apply():
    """size torch's intra-op pool to the budget (never above what torch chose itself); returns the thread count in force"""
    import torch

    n = min(cpu_budget(), torch.get_num_threads())
    if n != torch.get_num_threads():
        torch.set_num_threads(n)
    return torch.get_num_threads()
