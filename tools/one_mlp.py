"""single-kernel driver for PMC passes: the three fused ConvNeXt MLP kernels at the flagship's stage-0 shape (M = 262144, C = 96), 10 launches each
   python tools/one_mlp.py [C] [S]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K
torch.manual_seed(0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 96
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
M = 16 * S * S
bf = torch.bfloat16
y2 = torch.randn(M, C, device="cuda").to(bf)
dy = (torch.randn(M, C, device="cuda") * 0.1).to(bf)
W1 = torch.randn(C, 4 * C, device="cuda") / C ** 0.5
W2 = torch.randn(4 * C, C, device="cuda") / (4 * C) ** 0.5
b1 = torch.randn(4 * C, device="cuda") * 0.1
b2 = torch.randn(C, device="cuda") * 0.1
gamma = torch.rand(C, device="cuda") + 0.5
rs = torch.ones(16, device="cuda")
fw, bw = K.convnext_mlp_prep(W1, W2, gamma)
gW1, gb1, gW2, gb2, gg = (torch.zeros(s, device="cuda") for s in ((C, 4 * C), (4 * C,), (4 * C, C), (C,), (C,)))
res = torch.randn(M, C, device="cuda").to(bf)
for _ in range(10):
    K.convnext_mlp_fwd(y2, fw, b1, b2, gamma, rs, S * S, res)
    K.convnext_mlp_bwd_data(y2, dy, bw, b1, rs, S * S)
    K.convnext_mlp_wgrad(y2, dy, bw, b1, W2, b2, gamma, gW1, gb1, gW2, gb2, gg, rs, S * S)
torch.cuda.synchronize()
