"""Yardstick only (never on the product path): the library GEMM torch.matmul dispatches to, timed beside iseg_gemm on the
ConvNeXt-T flagship shapes (batch 16, 512x512).  Tells how far the hand-written kernel is from a tuned vendor kernel."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from iseg_amd import kernels as K  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print(f"{'shape':34s} {'ours us':>9s} {'lib us':>9s} {'ours TF/s':>10s} {'lib TF/s':>9s} {'min-bytes us':>12s}")
import os
SHAPES = ((96, 262144), (192, 65536), (384, 16384), (768, 4096)) if os.environ.get("KB_ALL", "1") == "1" else ((384, 16384), (768, 4096), (1024, 4096))
for C, M in SHAPES:
    x = torch.randn(M, C, device="cuda").bfloat16()
    h = torch.randn(M, 4 * C, device="cuda").bfloat16()
    w1 = torch.randn(C, 4 * C, device="cuda").bfloat16()      # [K][N]
    w2 = torch.randn(4 * C, C, device="cuda").bfloat16()
    y1 = torch.empty(M, 4 * C, device="cuda", dtype=torch.bfloat16)
    y2 = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    gw = torch.empty(C, 4 * C, device="cuda", dtype=torch.float32)
    cases = [
        ("pw1 fwd  NN", lambda: K.gemm(x, w1, y1, M, 4 * C, C, lda=C, ldb=4 * C, ldd=4 * C, a_kcontig=1, b_kcontig=0),
         lambda: torch.matmul(x, w1, out=y1), M, 4 * C, C, 2),
        ("pw2 fwd  NN", lambda: K.gemm(h, w2, y2, M, C, 4 * C, lda=4 * C, ldb=C, ldd=C, a_kcontig=1, b_kcontig=0),
         lambda: torch.matmul(h, w2, out=y2), M, C, 4 * C, 2),
        ("pw2 dgrad NT", lambda: K.gemm(x, w2, y1, M, 4 * C, C, lda=C, ldb=C, ldd=4 * C, a_kcontig=1, b_kcontig=1),
         lambda: torch.matmul(x, w2.t(), out=y1), M, 4 * C, C, 2),
        ("pw2 dgrad+gelu' NT", lambda: K.gemm(x, w2, y1, M, 4 * C, C, lda=C, ldb=C, ldd=4 * C, a_kcontig=1, b_kcontig=1, act=K.ACT_GELU_GRAD,
                                          aux=h, ldaux=4 * C),
         lambda: torch.matmul(x, w2.t(), out=y1), M, 4 * C, C, 2),
        ("pw2 dgrad+relu' NT", lambda: K.gemm(x, w2, y1, M, 4 * C, C, lda=C, ldb=C, ldd=4 * C, a_kcontig=1, b_kcontig=1, act=K.ACT_RELU_GRAD,
                                          aux=h, ldaux=4 * C),
         lambda: torch.matmul(x, w2.t(), out=y1), M, 4 * C, C, 2),
        ("pw1 fwd+gelu NN", lambda: K.gemm(x, w1, y1, M, 4 * C, C, lda=C, ldb=4 * C, ldd=4 * C, a_kcontig=1, b_kcontig=0, act=K.ACT_GELU),
         lambda: torch.matmul(x, w1, out=y1), M, 4 * C, C, 2),
        ("pw1 fwd+gelu+pre NN", lambda: K.gemm(x, w1, y1, M, 4 * C, C, lda=C, ldb=4 * C, ldd=4 * C, a_kcontig=1, b_kcontig=0, act=K.ACT_GELU,
                                              pre_out=h, ldp=4 * C),
         lambda: torch.matmul(x, w1, out=y1), M, 4 * C, C, 2),
        ("pw1 dgrad NT", lambda: K.gemm(h, w1, y2, M, C, 4 * C, lda=4 * C, ldb=4 * C, ldd=C, a_kcontig=1, b_kcontig=1),
         lambda: torch.matmul(h, w1.t(), out=y2), M, C, 4 * C, 2),
        ("pw1 wgrad TN", lambda: K.gemm(x, h, gw, C, 4 * C, M, lda=C, ldb=4 * C, ldd=4 * C, a_kcontig=0, b_kcontig=0),
         lambda: torch.matmul(x.t(), h), C, 4 * C, M, 4),
    ]
    for name, ours, lib, m, n, k, ob in cases:
        to, tl = timeit(ours), timeit(lib)
        fl = 2.0 * m * n * k
        by = (m * k + n * k) * 2 + m * n * ob
        print(f"{name} M={m} N={n} K={k}".ljust(34), f"{to:9.1f} {tl:9.1f} {fl / to / 1e6:10.1f} {fl / tl / 1e6:9.1f} {by / 8e6:12.1f}")
