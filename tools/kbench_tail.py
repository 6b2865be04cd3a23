#!/usr/bin/env python3
"""Logits tail of the flagship step (16 x 512 x 512, 21 classes, logits at 16 x 16): the fused upsample + CE + gradient + confusion kernel
against the three separate kernels.  usage: python3 tools/kbench_tail.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iseg_amd import kernels as K  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, Hi, Wi, C, Ho, Wo = 16, 16, 16, 21, 512, 512
z = (torch.randn(N, Hi, Wi, C, device="cuda") * 2).to(torch.bfloat16)
y = torch.randint(0, C, (N, Ho, Wo), dtype=torch.int32, device="cuda")
y[torch.rand(N, Ho, Wo, device="cuda") < 0.1] = 255
cm = torch.zeros(C * C, dtype=torch.int64, device="cuda")
P = N * Ho * Wo


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def separate():
    full = K.resize_bilinear(z, Ho, Wo, out_dtype=torch.float32)
    _, s, dl = K.softmax_ce_ignore(full.reshape(-1, C), y.reshape(-1), 255, want_px=False, want_sum=True, sum_scale=1.0 / P, want_grad=True,
                                   grad_scale=1.0 / P, cm=cm)
    return K.resize_bilinear_bwd(dl.reshape(N, Ho, Wo, C), Hi, Wi, torch.bfloat16)


tf = timeit(lambda: K.upsample_ce(z, y, Ho, Wo, 255, sum_scale=1.0 / P, grad_scale=1.0 / P, cm=cm))
ts = timeit(separate)
print(f"fused tail {tf:.1f} us ({(P * 4 + N * Hi * Wi * C * 4) / tf * 1e-3:.0f} GB/s of labels + logits), three kernels {ts:.1f} us")
