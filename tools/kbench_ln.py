"""LayerNorm forward / backward micro-benchmark at the four ConvNeXt-T stage shapes (16 images)"""
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
out = []
for S, C in [(128, 96), (64, 192), (32, 384), (16, 768)]:
    M = 16 * S * S
    x = torch.randn(M, C, device="cuda").bfloat16(); dy = torch.randn(M, C, device="cuda").bfloat16()
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6)
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    out.append(f"{S}x{C}: fwd {timeit(lambda: K.layernorm_fwd(x, g, b, 1e-6)):.1f} bwd {timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db)):.1f}")
print(" | ".join(out))
