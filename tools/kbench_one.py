#!/usr/bin/env python3
"""Run one family of kernels a few times at the flagship shapes (for rocprofv3 --kernel-trace / --pmc passes).
usage: python3 tools/kbench_one.py {dwbww|dwfwd|ln|gemm_s0} [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iseg_amd import kernels as K  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "dwbww"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shapes = [(128, 96), (64, 192), (32, 384), (16, 768)]
for S, C in shapes:
    x = torch.randn(16, S, S, C, device="cuda").to(torch.bfloat16)
    dy = torch.randn(16, S, S, C, device="cuda").to(torch.bfloat16)
    wd = torch.randn(49, C, device="cuda") / 7
    bd = torch.randn(C, device="cuda")
    dwg = torch.zeros(49, C, device="cuda")
    dbg = torch.zeros(C, device="cuda")
    for _ in range(iters):
        if what == "dwbww":
            K.dwconv2d_bwd_weight(x, dy, dwg, dbg, 7, 1, 3, 3)
        elif what == "dwfwd":
            K.dwconv2d(x, wd, bd, 7, 1, 3, 3)
        elif what == "ln":
            K.layernorm_fwd(x.reshape(-1, C), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), 1e-6)
        elif what == "gemm_s0":
            w = torch.randn(C, 4 * C, device="cuda").to(torch.bfloat16)
            pre = torch.empty(16 * S * S, 4 * C, device="cuda", dtype=torch.bfloat16)
            K.dense_fwd(x.reshape(-1, C), w, torch.zeros(4 * C, device="cuda"), act=K.ACT_GELU, pre_out=pre)
    torch.cuda.synchronize()
print("done", what)
