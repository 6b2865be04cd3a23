#!/usr/bin/env python3
"""the two weight-gradient products of an un-fused ConvNeXt block (stage 2 / 3 shapes): one by one (with the layer-scale slab consumer and the slab
reduce) against the pair launch of round 5"""
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
for rows, C in [(16384, 384), (4096, 768)]:
    bf = torch.bfloat16
    g = torch.randn(rows, 4 * C, device="cuda").to(bf); dbr = torch.randn(rows, C, device="cuda").to(bf)
    y2 = torch.randn(rows, C, device="cuda").to(bf); dh = torch.randn(rows, 4 * C, device="cuda").to(bf)
    W2 = torch.randn(4 * C, C, device="cuda"); b2 = torch.randn(C, device="cuda"); gamma = torch.rand(C, device="cuda")
    gW2, gg, gb2 = torch.zeros_like(W2), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    gW1, gb1 = torch.zeros(C, 4 * C, device="cuda"), torch.zeros(4 * C, device="cuda")
    def single():
        sl = K.dense_wgrad_slabs(g, dbr)
        K.layerscale_grads_slabs(sl[0], sl[1], W2, b2, gamma, gW2, gg, gb2)
        K.dense_wgrad(y2, dh, gW1, bias_grad=gb1)
    def pair():
        sl = K.dense_wgrad_pair(g, dbr, y2, dh, gW1, gb1)
        K.layerscale_grads_slabs(sl[0], sl[1], W2, b2, gamma, gW2, gg, gb2)
    if K.dense_wgrad_pair(g, dbr, y2, dh, gW1, gb1) is None:
        print(f"rows {rows} C {C}: one by one {timeit(single):.1f} us, pair: declined (too many tiles for two splits in one resident round)", flush=True)
        continue
    print(f"rows {rows} C {C}: one by one {timeit(single):.1f} us, pair {timeit(pair):.1f} us", flush=True)
