"""split-K sweep of the weight-gradient GEMM on the ConvNeXt-T shapes (slab writes + reduce included)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for C, M in ((96, 262144), (192, 65536), (384, 16384), (768, 4096)):
    for a, b in ((C, 4 * C), (4 * C, C)):
        x = torch.randn(M, a, device="cuda").bfloat16()
        dy = torch.randn(M, b, device="cuda").bfloat16()
        out = torch.zeros(a, b, device="cuda")
        res = []
        for split in [int(v) for v in os.environ.get("KB_SPLITS", "0,2,4,6,8,12,16,24,32,48").split(",")]:
            if split and M // split < 512:
                continue
            t = timeit(lambda: K.gemm(x, dy, out, a, b, M, lda=a, ldb=b, ldd=b, a_kcontig=0, b_kcontig=0, accumulate=True, split_k=split))
            res.append(f"{split}:{t:.1f}")
        print(f"wgrad [{a}x{b}] K={M}:  " + "  ".join(res))
