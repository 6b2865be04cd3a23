#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the cfg2 shapes (HIP events on the launch stream): achieved GB/s and TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from iseg_amd import kernels as K


def timeit(fn, iters=20, warm=3, name=""):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


ONLY = os.environ.get("KB_ONLY", "")


def report(name, sec, nbytes, flops=0):
    print(f"{name:58s} {sec * 1e6:9.1f} us  {nbytes / sec / 1e9:8.1f} GB/s  {flops / sec / 1e12:7.1f} TF/s", flush=True)


def main():
    B = int(os.environ.get("KB_BATCH", "16"))
    only = os.environ.get("KB_ONLY", "")
    dt = torch.bfloat16
    stages = [(128, 96), (64, 192), (32, 384), (16, 768)]
    for (S, C) in stages:
        M = B * S * S
        x = torch.randn(M, C, device="cuda").to(dt)
        w1 = (torch.randn(C, 4 * C, device="cuda") * C ** -0.5).to(dt)
        w2 = (torch.randn(4 * C, C, device="cuda") * (4 * C) ** -0.5).to(dt)
        b1 = torch.randn(4 * C, device="cuda")
        b2 = torch.randn(C, device="cuda")
        gam = torch.rand(C, device="cuda") + 0.5
        h = torch.empty(M, 4 * C, device="cuda", dtype=dt)
        g = torch.empty(M, 4 * C, device="cuda", dtype=dt)
        out = torch.empty(M, C, device="cuda", dtype=dt)
        t = timeit(lambda: K.dense_fwd(x, w1, b1, out=h))
        report(f"S{S} C{C} pw1 fwd (+bias -> h)", t, (M * C + M * 4 * C) * 2, 2 * M * C * 4 * C)
        t = timeit(lambda: K.dense_fwd(h, w2, b2, colscale=gam, residual=x, out=out, a_act=K.ACT_GELU))
        report(f"S{S} C{C} pw2 fwd (gelu(A)+bias+scale+residual)", t, (M * 4 * C + 2 * M * C) * 2, 2 * M * C * 4 * C)
        dh = torch.empty(M, 4 * C, device="cuda", dtype=dt)
        t = timeit(lambda: K.dense_dgrad(out, w2, act=K.ACT_GELU_GRAD, aux=h, out=dh))
        report(f"S{S} C{C} pw2 dgrad (*gelu'(h))", t, (M * C + 2 * M * 4 * C) * 2, 2 * M * C * 4 * C)
        t = timeit(lambda: K.dense_dgrad(dh, w1, out=out))
        report(f"S{S} C{C} pw1 dgrad", t, (M * 4 * C + M * C) * 2, 2 * M * C * 4 * C)
        Z = torch.empty(4 * C, C, device="cuda")
        t = timeit(lambda: K.dense_wgrad(h, out, Z, accumulate=False, a_act=K.ACT_GELU))
        report(f"S{S} C{C} pw2 wgrad (gelu(h)^T dout)", t, (M * 4 * C + M * C) * 2, 2 * M * C * 4 * C)
        dW1 = torch.zeros(C, 4 * C, device="cuda")
        t = timeit(lambda: K.dense_wgrad(x, dh, dW1))
        report(f"S{S} C{C} pw1 wgrad (y2^T dh)", t, (M * 4 * C + M * C) * 2, 2 * M * C * 4 * C)
        s4 = torch.empty(4 * C, device="cuda")
        t = timeit(lambda: K.colsum(dh, 4 * C, 0, 1, M, 4 * C, s4))
        report(f"S{S} C{C} colsum(dh)", t, M * 4 * C * 2)
        x4 = x.reshape(B, S, S, C)
        wd = torch.randn(49, C, device="cuda") / 7
        bd = torch.randn(C, device="cuda")
        t = timeit(lambda: K.dwconv2d(x4, wd, bd, 7, 1, 3, 3))
        report(f"S{S} C{C} dwconv7 fwd", t, 2 * M * C * 2, 2 * 49 * M * C)
        t = timeit(lambda: K.dwconv2d(x4, wd, None, 7, 1, 3, 3, flip=True, add=x4))
        report(f"S{S} C{C} dwconv7 bwd-data (+add)", t, 3 * M * C * 2, 2 * 49 * M * C)
        dwg = torch.zeros(49, C, device="cuda")
        dbg = torch.zeros(C, device="cuda")
        t = timeit(lambda: K.dwconv2d_bwd_weight(x4, x4, dwg, dbg, 7, 1, 3, 3))
        report(f"S{S} C{C} dwconv7 bwd-weight", t, 2 * M * C * 2, 2 * 49 * M * C)
        ga, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        t = timeit(lambda: K.layernorm_fwd(x, ga, be, 1e-6))
        report(f"S{S} C{C} layernorm fwd", t, 2 * M * C * 2)
        y, mean, rstd = K.layernorm_fwd(x, ga, be, 1e-6)
        dga, dbe = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        t = timeit(lambda: K.layernorm_bwd(x, x, ga, mean, rstd, dga, dbe))
        report(f"S{S} C{C} layernorm bwd", t, 3 * M * C * 2)
        del x, h, g, dh, out
    P = B * 512 * 512
    z = torch.randn(P, 21, device="cuda")
    y = torch.randint(0, 21, (P,), device="cuda", dtype=torch.int32)
    t = timeit(lambda: K.softmax_ce_ignore(z, y, 255, want_px=False, want_sum=True, want_grad=True))
    report("softmax CE fwd+bwd [P,21]", t, 2 * P * 21 * 4)
    cm = torch.zeros(441, dtype=torch.int64, device="cuda")
    t = timeit(lambda: K.argmax_confusion(z, y, 255, cm=cm))
    report("argmax+confusion [P,21]", t, P * 21 * 4)
    small = torch.randn(B, 16, 16, 21, device="cuda").to(dt)
    t = timeit(lambda: K.resize_bilinear(small, 512, 512, out_dtype=torch.float32))
    report("resize 16->512 fwd", t, P * 21 * 4)
    dz = z.reshape(B, 512, 512, 21)
    t = timeit(lambda: K.resize_bilinear_bwd(dz, 16, 16, dt))
    report("resize 16->512 bwd", t, P * 21 * 4)


if __name__ == "__main__":
    main()
