#!/usr/bin/env python3
"""GEMM variants A/B on the stage-0/1 shapes (env knobs are read once per process)."""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath("%s"))))
from iseg_amd import kernels as K
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out=[]
dt=torch.bfloat16
for (S, C) in [(128, 96), (64, 192), (32, 384), (16, 768)]:
    M = 16*S*S
    x = torch.randn(M, C, device="cuda").to(dt)
    w1 = (torch.randn(C, 4*C, device="cuda") * C**-0.5).to(dt); w2 = (torch.randn(4*C, C, device="cuda") * (4*C)**-0.5).to(dt)
    b1 = torch.randn(4*C, device="cuda"); b2 = torch.randn(C, device="cuda"); gam = torch.rand(C, device="cuda")+0.5
    h = torch.empty(M, 4*C, device="cuda", dtype=dt); g = torch.empty(M, 4*C, device="cuda", dtype=dt); o = torch.empty(M, C, device="cuda", dtype=dt)
    dh = torch.empty(M, 4*C, device="cuda", dtype=dt)
    a = timeit(lambda: K.dense_fwd(x, w1, b1, act=K.ACT_GELU, pre_out=h, out=g))
    b = timeit(lambda: K.dense_fwd(g, w2, b2, colscale=gam, residual=x, out=o))
    c = timeit(lambda: K.dense_dgrad(o, w2, act=K.ACT_GELU_GRAD, aux=h, out=dh))
    d = timeit(lambda: K.dense_dgrad(dh, w1, out=o))
    Z = torch.empty(4*C, C, device="cuda")
    e = timeit(lambda: K.dense_wgrad(g, o, Z, accumulate=False))
    dW1 = torch.zeros(C, 4*C, device="cuda")
    db1 = torch.zeros(4*C, device="cuda")
    f = timeit(lambda: K.dense_wgrad(x, dh, dW1, bias_grad=db1))
    out.append(f"C{C}: pw1f {a:5.0f} pw2f {b:5.0f} dg2 {c:5.0f} dg1 {d:5.0f} wg2 {e:5.0f} wg1 {f:5.0f}")
print(os.environ.get("TAG"), " | ".join(out), flush=True)
''' % os.path.abspath(__file__)
for bk in ("64", "128"):
    env = dict(os.environ, ISEG_GEMM_BK=bk, TAG=f"bk={bk}")
    subprocess.run([sys.executable, "-c", code], env=env)
