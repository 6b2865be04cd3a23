#!/usr/bin/env python3
"""Strided convolution data gradient: stride phases in one implicit-GEMM launch (csrc/conv_igemm.hip pass 3) against GEMM + col2im, at the
flagship's downsample shapes and ResNet-50's strided 3x3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K

BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (S, Cin, Cout, k, s) in [(128, 96, 192, 2, 2), (64, 192, 384, 2, 2), (32, 384, 768, 2, 2), (64, 128, 128, 3, 2), (32, 256, 256, 3, 2), (64, 256, 512, 1, 2)]:
    N = 16
    Ho, pt = K.same_pad(S, k, s, 1)
    geom = K.conv_geom(N, S, S, Cin, Cout, k, k, s, s, 1, 1, pt, pt, Ho, Ho, 1)
    w = (torch.randn(k, k, Cin, Cout, device="cuda") * 0.05).to(BF)
    dy = torch.randn(N, Ho, Ho, Cout, device="cuda").to(BF)
    M, Kd = N * Ho * Ho, k * k * Cin
    ldc = (Kd + 7) // 8 * 8

    def col_route():
        dcol = torch.empty((M, ldc), dtype=BF, device="cuda")
        K.gemm(dy.reshape(M, Cout), w.reshape(Kd, Cout), dcol, M, Kd, Cout, lda=Cout, ldb=Cout, ldd=ldc, a_kcontig=1, b_kcontig=1)
        return K.col2im(dcol, N, S, S, Cin, k, k, s, s, 1, 1, pt, pt, Ho, Ho)

    a, b = K.conv2d_igemm_bwd_data(dy, w, geom), col_route()
    err = (a.float() - b.float()).abs().max().item()
    print(f"{S}x{S}x{Cin}->{Cout} k{k}s{s}: phases {timeit(lambda: K.conv2d_igemm_bwd_data(dy, w, geom)):6.1f} us   gemm+col2im {timeit(col_route):6.1f} us   max|diff| {err:.3g}",
          flush=True)
