"""single-kernel driver for PMC passes: pwconv1 forward of a stage-2 ConvNeXt block (gelu + gelu' outputs), 20 launches"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K
torch.manual_seed(0)
M, C, H = 16384, 384, 1536
x = (torch.randn(M, C, device="cuda") * 0.5).to(torch.bfloat16)
w1t = (torch.randn(H, C, device="cuda") * C ** -0.5).to(torch.bfloat16)
b1 = torch.randn(H, device="cuda") * 0.1
o1 = torch.empty(M, H, dtype=torch.bfloat16, device="cuda"); p1 = torch.empty_like(o1)
for _ in range(20):
    K.dense_fwd_t(x, w1t, b1, act=K.ACT_GELU, out=o1, pre_out=p1, pre_deriv=True)
torch.cuda.synchronize()
