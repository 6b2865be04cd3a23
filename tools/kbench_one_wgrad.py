#!/usr/bin/env python3
"""a few launches of the fused-MLP weight-gradient kernel alone (for tools/pmc_kernel.sh): python3 tools/kbench_one_wgrad.py [C]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iseg_amd import kernels as K  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 96
S = 128 if C == 96 else 64
M = 16 * S * S
bf = torch.bfloat16
y2 = torch.randn(M, C, device="cuda").to(bf)
dy = torch.randn(M, C, device="cuda").to(bf)
W1 = torch.randn(C, 4 * C, device="cuda") / C ** 0.5
W2 = torch.randn(4 * C, C, device="cuda") / (4 * C) ** 0.5
b1 = torch.randn(4 * C, device="cuda") * 0.1
b2 = torch.randn(C, device="cuda") * 0.1
gamma = torch.rand(C, device="cuda") + 0.5
rs = torch.ones(16, device="cuda")
fw, bw = K.convnext_mlp_prep(W1, W2, gamma)
g = [torch.zeros(s, device="cuda") for s in ((C, 4 * C), (4 * C,), (4 * C, C), (C,), (C,))]
for _ in range(4):
    K.convnext_mlp_wgrad(y2, dy, bw, b1, W2, b2, gamma, *g, rs, S * S)
    K.convnext_mlp_bwd_data(y2, dy, bw, b1, rs, S * S)
torch.cuda.synchronize()
