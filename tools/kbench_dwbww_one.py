import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from iseg_amd import kernels as K
S, C = 128, 96
x = torch.randn(16, S, S, C, device="cuda").to(torch.bfloat16)
dwg = torch.zeros(49, C, device="cuda"); dbg = torch.zeros(C, device="cuda")
for _ in range(5):
    K.dwconv2d_bwd_weight(x, x, dwg, dbg, 7, 1, 3, 3)
torch.cuda.synchronize()
