"""Yardstick only (never on the product path): which library kernels torch.matmul dispatches to on the stage-2 / 3 data-path shapes -- run under
rocprofv3 --kernel-trace (tools/ktrace.sh); the kernel names carry the library's macro tile, split strategy and staging."""
import torch

for M, N, Kd in ((16384, 384, 1536), (4096, 768, 3072), (16384, 1536, 384), (4096, 3072, 768)):
    a = torch.randn(M, Kd, device="cuda").bfloat16()
    w = torch.randn(N, Kd, device="cuda").bfloat16()      # K-contiguous, as our weight copies are
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(12):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
