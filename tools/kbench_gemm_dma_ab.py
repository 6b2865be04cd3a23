#!/usr/bin/env python3
"""One LDS-DMA GEMM shape, timed with the environment's tile knobs (run once per setting: the knobs are read at first use).
  python tools/kbench_gemm_dma_ab.py M N K [aux]      # aux: the x aux epilogue (dgrad x gelu')"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import kernels as K  # noqa: E402

M, N, Kd = (int(v) for v in sys.argv[1:4])
aux_on = len(sys.argv) > 4
torch.manual_seed(0)
a = torch.randn(M, Kd, device="cuda").bfloat16()
b = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).bfloat16()
aux = torch.rand(M, N, device="cuda").bfloat16() if aux_on else None
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)


def run():
    K.gemm(a, b, out, M, N, Kd, lda=Kd, ldb=Kd, ldd=N, a_kcontig=1, b_kcontig=1, act=K.ACT_MUL_AUX if aux_on else K.ACT_NONE, aux=aux,
           ldaux=N if aux_on else 0)


run()
ref = a.float() @ b.float().t()
if aux_on:
    ref = ref * aux.float()
err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print(f"M={M} N={N} K={Kd} aux={aux_on}  {us:7.1f} us  {2.0 * M * N * Kd / us * 1e-6:7.1f} TFLOP/s  rel err {err:.2e}  "
      f"WIDE={os.environ.get('ISEG_GEMM_DMA_WIDE', '0')} PERSIST={os.environ.get('ISEG_GEMM_DMA_PERSIST', '1')}")
