#!/usr/bin/env python3
"""Backward pass of the fused ConvNeXt MLP at the flagship stage-0 / stage-1 shapes: the round-2 route (chain kernel writing g / dh + two
weight-gradient GEMMs + layer-scale helper) against the round-3 route (data-gradient chain + recomputing weight-gradient kernel).
usage: python3 tools/kbench_mlp_bwd.py [iters]"""
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
    dy = torch.randn(M, C, device="cuda").to(bf)
    W1 = torch.randn(C, 4 * C, device="cuda") / C ** 0.5
    W2 = torch.randn(4 * C, C, device="cuda") / (4 * C) ** 0.5
    b1 = torch.randn(4 * C, device="cuda") * 0.1
    b2 = torch.randn(C, device="cuda") * 0.1
    gamma = torch.rand(C, device="cuda") + 0.5
    rs = torch.ones(16, device="cuda")
    fw, bw = K.convnext_mlp_prep(W1, W2, gamma)
    gW1, gb1, gW2, gb2, gg = (torch.zeros(s, device="cuda") for s in ((C, 4 * C), (4 * C,), (4 * C, C), (C,), (C,)))
    Z = torch.empty(4 * C, C, device="cuda")
    Ssum = torch.empty(C, device="cuda")

    def old():
        dbr = K.rowscale(dy, rs, S * S)
        K.colsum(dbr, C, 0, 1, M, C, Ssum)
        g, dh, dy2 = K.convnext_mlp_bwd(y2, dbr, bw, b1)
        K.dense_wgrad(g, dbr, Z, accumulate=False)
        K.layerscale_grads(Z, W2, b2, gamma, Ssum, gW2, gg, gb2)
        K.dense_wgrad(y2, dh, gW1, bias_grad=gb1)
        return dy2

    t_old = timeit(old)
    t_data = timeit(lambda: K.convnext_mlp_bwd_data(y2, dy, bw, b1, rs, S * S))
    t_wg = timeit(lambda: K.convnext_mlp_wgrad(y2, dy, bw, b1, W2, b2, gamma, gW1, gb1, gW2, gb2, gg, rs, S * S))
    print(f"C={C} M={M}: round-2 route {t_old:.1f} us | data-gradient chain {t_data:.1f} us + weight gradients {t_wg:.1f} us = {t_data + t_wg:.1f} us")
