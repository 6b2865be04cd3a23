"""single-process driver for PMC passes: the four data-path GEMMs of a stage-2 ConvNeXt block (16 x 32 x 32 pixels, C = 384), 10 launches each:
pwconv1 forward (gelu + gelu'), pwconv2 forward (bias, residual), pwconv2 data gradient (x aux), pwconv1 data gradient (plain)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K
torch.manual_seed(0)
M, C, H = 16384, 384, 1536
x = (torch.randn(M, C, device="cuda") * 0.5).to(torch.bfloat16)
r = (torch.randn(M, C, device="cuda") * 0.5).to(torch.bfloat16)
w1t = (torch.randn(H, C, device="cuda") * C ** -0.5).to(torch.bfloat16)
w2t = (torch.randn(C, H, device="cuda") * H ** -0.5).to(torch.bfloat16)
b1 = torch.randn(H, device="cuda") * 0.1
b2 = torch.randn(C, device="cuda") * 0.1
g = torch.empty(M, H, dtype=torch.bfloat16, device="cuda"); h = torch.empty_like(g); dh = torch.empty_like(g)
o = torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
for _ in range(10):
    K.gemm(x, w1t, g, M, H, C, lda=C, ldb=C, ldd=H, a_kcontig=1, b_kcontig=1, bias=b1, act=K.ACT_GELU, pre_out=h, ldp=H, pre_deriv=True)
    K.gemm(g, w2t, o, M, C, H, lda=H, ldb=H, ldd=C, a_kcontig=1, b_kcontig=1, bias=b2, residual=r, ldr=C)
    K.gemm(x, w1t, dh, M, H, C, lda=C, ldb=C, ldd=H, a_kcontig=1, b_kcontig=1, act=K.ACT_MUL_AUX, aux=h, ldaux=H)
    K.gemm(dh, w2t, o, M, C, H, lda=H, ldb=H, ldd=C, a_kcontig=1, b_kcontig=1)
torch.cuda.synchronize()
