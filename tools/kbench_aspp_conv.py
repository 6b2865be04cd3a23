#!/usr/bin/env python3
"""The flagship's ASPP 3x3 dilated convolution (16 x 16 x 16 x 768 -> 256, dilation 6) through the implicit-GEMM entry points: forward, data
gradient, weight gradient (us per call, incl. the split-K slab sums)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K

BF = torch.bfloat16


def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


N, S, Cin, Cout, k, d = 16, 16, 768, 256, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 6
Ho, pt = K.same_pad(S, k, 1, d)
geom = K.conv_geom(N, S, S, Cin, Cout, k, k, 1, 1, d, d, pt, pt, Ho, Ho, 1)
x = torch.randn(N, S, S, Cin, device="cuda").to(BF)
w = (torch.randn(k, k, Cin, Cout, device="cuda") * 0.02).to(BF)
dy = torch.randn(N, Ho, Ho, Cout, device="cuda").to(BF)
dw = torch.zeros(k, k, Cin, Cout, device="cuda")
wt = w.reshape(-1, Cout).t().contiguous()
kt = f"{timeit(lambda: K.conv2d_igemm_fwd_kt(x, wt, None, geom)):.1f}" if K.conv2d_igemm_fwd_kt_supported(geom, BF) else "-"
print(f"fwd {timeit(lambda: K.conv2d_igemm_fwd(x, w, None, geom)):.1f} us (LDS-DMA on the K-contiguous kernel: {kt}) | bwd data {timeit(lambda: K.conv2d_igemm_bwd_data(dy, w, geom)):.1f} us | "
      f"bwd weight {timeit(lambda: K.conv2d_igemm_bwd_weight(x, dy, dw, geom, accumulate=True)):.1f} us", flush=True)
