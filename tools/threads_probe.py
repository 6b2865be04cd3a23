import os, time, sys
sys.path.insert(0, "/root/repo")
import torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "-", type(e).__name__)
import torch.nn.functional as F
def work(dtype):
    torch.manual_seed(0)
    x = torch.randn(2, 96, 128, 128, dtype=dtype, requires_grad=True)
    w = torch.randn(96, 1, 7, 7, dtype=dtype, requires_grad=True)
    w2 = torch.randn(384, 96, 1, 1, dtype=dtype, requires_grad=True)
    t0 = time.perf_counter()
    for _ in range(3):
        y = F.conv2d(x, w, padding=3, groups=96)
        y = F.layer_norm(y.permute(0, 2, 3, 1), (96,)).permute(0, 3, 1, 2)
        y = F.gelu(F.conv2d(y, w2))
        y.sum().backward()
    return (time.perf_counter() - t0) / 3
for n in (None, 64, 32, 16, 8):
    if n:
        torch.set_num_threads(n)
    print("threads", torch.get_num_threads(), "fp64 %.3f s" % work(torch.float64), "fp32 %.3f s" % work(torch.float32), flush=True)
