#!/usr/bin/env python3
"""Fused ConvNeXt MLP (csrc/mlp_fused.hip) against the un-fused GEMM pair at the flagship stage-0 / stage-1 shapes.
usage: python3 tools/kbench_mlp.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iseg_amd import kernels as K  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20


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


for S, C in [(128, 96), (64, 192)]:
    M = 16 * S * S
    bf = torch.bfloat16
    y2 = torch.randn(M, C, device="cuda").to(bf)
    res = torch.randn(M, C, device="cuda").to(bf)
    dy = torch.randn(M, C, device="cuda").to(bf)
    W1 = (torch.randn(C, 4 * C, device="cuda") / C ** 0.5).to(bf)
    W2 = (torch.randn(4 * C, C, device="cuda") / (4 * C) ** 0.5).to(bf)
    b1 = torch.randn(4 * C, device="cuda") * 0.1
    b2 = torch.randn(C, device="cuda") * 0.1
    gamma = torch.rand(C, device="cuda") + 0.5
    rs = torch.ones(16, device="cuda")
    pre = torch.empty(M, 4 * C, device="cuda", dtype=bf)

    def unfused():
        g = K.dense_fwd(y2, W1, b1, act=K.ACT_GELU, pre_out=pre, pre_deriv=True)
        return K.dense_fwd(g, W2, b2, colscale=gamma, rowscale=rs, rows_per_group=S * S, residual=res)

    W1f, W2f = W1.float(), W2.float()
    fw, bw = K.convnext_mlp_prep(W1f, W2f, gamma)

    def fused():
        return K.convnext_mlp_fwd(y2, fw, b1, b2, gamma, rs, S * S, res)

    tu, tf = timeit(unfused), timeit(fused)
    flops = 2 * 2 * M * C * 4 * C
    print(f"C={C} M={M}: unfused fwd {tu:.1f} us, fused fwd {tf:.1f} us ({flops / tf * 1e-6:.0f} TFLOP/s, {4 * M * C * 2 / tf * 1e-3:.0f} GB/s algorithmic)")
    tp = timeit(lambda: K.convnext_mlp_prep(W1f, W2f, gamma))
    tb = timeit(lambda: K.convnext_mlp_bwd(y2, dy, bw, b1))

    def unfused_bwd():
        dh = K.dense_dgrad(dy, W2, act=K.ACT_MUL_AUX, aux=pre)
        return K.dense_dgrad(dh, W1)

    tub = timeit(unfused_bwd)
    print(f"          prep {tp:.1f} us, fused bwd chain {tb:.1f} us ({(4 * M * C * 2 + 2 * M * 4 * C * 2 + M * C * 2) / tb * 1e-3:.0f} GB/s algorithmic), "
          f"unfused dgrad pair {tub:.1f} us")
