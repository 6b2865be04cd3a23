"""DCNv3 backward at the InternImage-B stage shapes (batch 8, 512 x 512 input): python tools/kbench_dcn.py"""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath("%s"))))
from iseg_amd import kernels as K
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = []
for (S, C) in [(128, 112), (64, 224), (32, 448), (16, 896)]:
    G = C // 16
    x = torch.randn(8, S, S, C, device="cuda").to(torch.bfloat16)
    off = (torch.randn(8, S, S, G * 18, device="cuda") * float(os.environ.get("SPREAD", "0.5"))).to(torch.bfloat16)
    m = torch.softmax(torch.randn(8, S, S, G, 9, device="cuda"), -1).reshape(8, S, S, G * 9).to(torch.bfloat16)
    f = timeit(lambda: K.dcnv3_fwd(x, off, m, G, 16, 3, 3, 1, 1, 1, 1.0))
    b = timeit(lambda: K.dcnv3_bwd(x, off, m, x, G, 16, 3, 3, 1, 1, 1, 1.0))
    # the layer's route since round 5: offsets | mask as column ranges of one matrix, dx in bf16, side buffer kept zero (K.dcnv3_bwd_joint)
    gp = G * 9
    om = torch.zeros(8 * S * S, (3 * gp + 7) // 8 * 8, device="cuda", dtype=torch.bfloat16)
    om[:, :2 * gp] = off.reshape(-1, 2 * gp)
    om[:, 2 * gp:3 * gp] = m.reshape(-1, gp)
    bj = timeit(lambda: K.dcnv3_bwd_joint(x, om, x, G, 16, 3, 3, 1, 1, 1, 1.0))
    out.append(f"S{S}C{C} fwd {f:7.1f} bwd {b:7.1f} joint bwd {bj:7.1f} ({bj / f:.2f}x)")
print(os.environ.get("TAG"), " | ".join(out), flush=True)
''' % os.path.abspath(__file__)
for win in ("1", "0"):
    for spread in ("0.5", "4.0"):
        env = dict(os.environ, ISEG_DCN_BWD_WIN=win, SPREAD=spread, TAG=f"win={win} spread={spread}")
        subprocess.run([sys.executable, "-c", code], env=env)
