"""micro-benchmark of the inference attention kernel on the ViT-B/16 window shape (4 windows x 1025 tokens x 12 heads x 64)"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from iseg_amd import kernels as K  # noqa: E402

B, T, H, D = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (4, 1025, 12, 64)))
qkv = (torch.randn(B, T, 3 * H * D, device="cuda") * 1.0).bfloat16()
for _ in range(5):
    K.attention_fwd(qkv, H, D ** -0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    K.attention_fwd(qkv, H, D ** -0.5)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
flops = 4.0 * B * H * T * T * D
print(f"flash fwd B={B} T={T} H={H}: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s")
