"""race screen for the LDS-DMA GEMM pipeline: many back-to-back launches of every tile variant against an fp32 reference product
(a stale or early-read stage would show up as grossly wrong tiles, not as rounding noise)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K  # noqa: E402

torch.manual_seed(0)
shapes = [(300, 64, 256), (1000, 136, 128), (7000, 384, 192), (16384, 512, 128), (25000, 264, 320), (640, 128, 2048), (16384, 1536, 384),
          (4096, 768, 3072), (40000, 640, 128), (33000, 1032, 448), (70000, 768, 192), (4096, 3072, 768), (66000, 576, 128)]      # several 256 x 128 tiles per CU (persistent form), ragged; N % 192 == 0 with >= 256 tiles (256 x 192 tiles)
worst = 0.0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for M, N, Kd in shapes:
        a = torch.randn(M, Kd, device="cuda").bfloat16()
        b = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        K.gemm(a, b, out, M, N, Kd, lda=Kd, ldb=Kd, ldd=N, a_kcontig=1, b_kcontig=1)
        ref = a.float() @ b.float().t()
        err = (out.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        worst = max(worst, err)
        if err > 2e-2:
            print("MISMATCH", it, (M, N, Kd), err)
            sys.exit(1)
print("ok, worst relative error", worst)
