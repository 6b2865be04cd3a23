#!/usr/bin/env python3
"""the matrix-core depthwise 7 x 7 (csrc/dwconv_mfma.hip) at the four flagship stage shapes: forward and data gradient (flip + residual add)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K
def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = []
for (S, C) in [(128, 96), (64, 192), (32, 384), (16, 768)]:
    x = torch.randn(16, S, S, C, device="cuda").to(torch.bfloat16)
    wd = torch.randn(49, C, device="cuda") / 7; bd = torch.randn(C, device="cuda")
    f = timeit(lambda: K.dwconv2d7_mfma(x, wd, bd))
    g = timeit(lambda: K.dwconv2d7_mfma(x, wd, None, flip=True, add=x))
    out.append(f"S{S}C{C} fwd {f:6.1f} bwd-data {g:6.1f}")
print(sys.argv[1] if len(sys.argv) > 1 else "", " | ".join(out), flush=True)
