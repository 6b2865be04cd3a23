#!/usr/bin/env python3
"""Layer-scale gradients of one un-fused ConvNeXt block (stage 2: 1536 x 384): weight-gradient product + slab sum + iseg_layerscale_grads against
the product stopped at its slabs + iseg_layerscale_grads_slabs (us per route, incl. the product)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K


def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, Kd, Nd) in [(16384, 1536, 384), (4096, 3072, 768)]:
    g = (torch.randn(M, Kd, device="cuda") * 0.5).to(torch.bfloat16)
    dbr = (torch.randn(M, Nd, device="cuda") * 0.1).to(torch.bfloat16)
    W2, b2, gamma = torch.randn(Kd, Nd, device="cuda") * 0.05, torch.randn(Nd, device="cuda") * 0.1, torch.rand(Nd, device="cuda") + 0.5
    dW2, dg, db = torch.zeros(Kd, Nd, device="cuda"), torch.zeros(Nd, device="cuda"), torch.zeros(Nd, device="cuda")
    Z, S = torch.empty(Kd, Nd, device="cuda"), torch.empty(Nd, device="cuda")

    def tensor_route():
        K.dense_wgrad(g, dbr, Z, accumulate=False, bias_grad=S)
        K.layerscale_grads(Z, W2, b2, gamma, S, dW2, dg, db)

    def slab_route():
        sl = K.dense_wgrad_slabs(g, dbr)
        K.layerscale_grads_slabs(sl[0], sl[1], W2, b2, gamma, dW2, dg, db)

    def product_only():
        K.dense_wgrad_slabs(g, dbr)

    print(f"M={M} {Kd}x{Nd}: slab sum + Z + layer scale {timeit(tensor_route):.1f} us | from the slabs {timeit(slab_route):.1f} us | product alone {timeit(product_only):.1f} us",
          flush=True)
