#!/usr/bin/env python3
"""dwconv variants A/B (env knobs are read once per process, so each variant runs in its own process)."""
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
for (S, C) in [(128, 96), (64, 192), (32, 384), (16, 768)]:
    x = torch.randn(16, S, S, C, device="cuda").to(torch.bfloat16)
    wd = torch.randn(49, C, device="cuda") / 7; bd = torch.randn(C, device="cuda")
    dwg = torch.zeros(49, C, device="cuda"); dbg = torch.zeros(C, device="cuda")
    f = timeit(lambda: K.dwconv2d(x, wd, bd, 7, 1, 3, 3))
    g = timeit(lambda: K.dwconv2d(x, wd, None, 7, 1, 3, 3, flip=True, add=x))
    b = timeit(lambda: K.dwconv2d_bwd_weight(x, x, dwg, dbg, 7, 1, 3, 3))
    out.append(f"S{S}C{C} fwd {f:6.1f} bwd-data {g:6.1f} bww {b:6.1f}")
print(os.environ.get("TAG"), " | ".join(out), flush=True)
''' % os.path.abspath(__file__)
subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ISEG_DW_MFMA="1", TAG="mfma (dwconv_mfma.hip)"))
for dma in ("1", "0"):
    env = dict(os.environ, ISEG_DW_MFMA="0", ISEG_DW_BW_DMA=dma, ISEG_DW_FWD_DMA=dma, TAG=f"valu dma={dma}")
    subprocess.run([sys.executable, "-c", code], env=env)
for slots in sys.argv[1:]:      # extra arguments: resident-workgroup targets of the DMA-tiled weight gradient (ISEG_DW_BW_DMA_SLOTS)
    env = dict(os.environ, ISEG_DW_BW_DMA_SLOTS=slots, TAG=f"dma=1 bw slots={slots}")
    subprocess.run([sys.executable, "-c", code], env=env)
