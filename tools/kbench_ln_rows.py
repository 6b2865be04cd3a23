"""LayerNorm backward / forward at growing row counts (fixed cost vs streaming rate); run under rocprofv3 --kernel-trace --stats for kernel-only times"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from iseg_amd import kernels as K
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for C in (384, 768, 96):
    out = []
    for M in (256, 1024, 4096, 16384, 65536, 262144):
        if M * C > 262144 * 96 * 2: continue
        x = torch.randn(M, C, device="cuda").bfloat16(); dy = torch.randn(M, C, device="cuda").bfloat16()
        g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6)
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        out.append(f"M={M}: fwd {timeit(lambda: K.layernorm_fwd(x, g, b, 1e-6)):.1f} bwd {timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db)):.1f}")
    print(f"C={C}: " + " | ".join(out), flush=True)
