"""DCNv3 forward / backward at the stride-4 shape of InternImage-B (8 x 128 x 128 x 112), one line: python tools/kbench_dcn_s128.py [tag]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


S, C = 128, 112
G = C // 16
x = torch.randn(8, S, S, C, device="cuda").to(torch.bfloat16)
off = (torch.randn(8, S, S, G * 18, device="cuda") * 0.5).to(torch.bfloat16)
m = torch.softmax(torch.randn(8, S, S, G, 9, device="cuda"), -1).reshape(8, S, S, G * 9).to(torch.bfloat16)
f = timeit(lambda: K.dcnv3_fwd(x, off, m, G, 16, 3, 3, 1, 1, 1, 1.0))
b = timeit(lambda: K.dcnv3_bwd(x, off, m, x, G, 16, 3, 3, 1, 1, 1, 1.0))
print(sys.argv[1] if len(sys.argv) > 1 else "", f"S{S}C{C} fwd {f:7.1f} bwd {b:7.1f}", flush=True)
