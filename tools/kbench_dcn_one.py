"""One DCNv3 forward + backward shape in-process (for rocprofv3 --kernel-trace --stats): python tools/kbench_dcn_one.py [S] [C] [spread]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K

S, C = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 112
spread = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
G = C // 16
x = torch.randn(8, S, S, C, device="cuda").to(torch.bfloat16)
off = (torch.randn(8, S, S, G * 18, device="cuda") * spread).to(torch.bfloat16)
m = torch.softmax(torch.randn(8, S, S, G, 9, device="cuda"), -1).reshape(8, S, S, G * 9).to(torch.bfloat16)
for _ in range(12):
    K.dcnv3_fwd(x, off, m, G, 16, 3, 3, 1, 1, 1, 1.0)
    K.dcnv3_bwd(x, off, m, x, G, 16, 3, 3, 1, 1, 1, 1.0)
torch.cuda.synchronize()
