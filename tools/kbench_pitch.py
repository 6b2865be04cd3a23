"""Row-pitch experiment for the stage-2 / stage-3 LDS-DMA GEMMs: the same products with the K-contiguous operands and the outputs
at their dense pitch (K or N elements: 768 / 3072 / 6144 bytes, multiples of 256 B whose rows fall on a few L2 channels) and at a
padded pitch (+ PAD elements).  us per launch, 20 launches back to back."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def padded(rows, cols, pad):
    """[rows, cols] view of a [rows, cols + pad] bf16 buffer"""
    buf = torch.randn(rows, cols + pad, device="cuda").bfloat16()
    return buf[:, :cols], cols + pad


PADS = [int(p) for p in os.environ.get("KB_PADS", "0,64,128,32").split(",")]
print("pads (elements):", PADS)
for C, M in ((384, int(os.environ.get("KB_M2", "16384"))), (768, int(os.environ.get("KB_M3", "4096")))):
    H = 4 * C
    bias = torch.randn(H, device="cuda")
    bias_c = torch.randn(C, device="cuda")
    for name in ("pw1 fwd gelu+deriv", "pw2 fwd bias+res", "dgrad-pw2 mul aux", "dgrad-pw1 plain"):
        line = f"C={C} M={M} {name:20s}"
        for pa in PADS:          # pitch pad of every [M, 4C]-shaped tensor (hidden activations)
            for pw in (0, pa) if pa else (0,):      # and of the K-contiguous weight copies
                x, ldx = padded(M, C, 0)
                if name == "pw1 fwd gelu+deriv":      # g = gelu(x W1t^T + b), h = gelu'
                    w, ldw = padded(H, C, 0)
                    g, ldg = padded(M, H, pa)
                    h, ldh = padded(M, H, pa)
                    fn = lambda: K.gemm(x, w, g, M, H, C, lda=ldx, ldb=ldw, ldd=ldg, a_kcontig=1, b_kcontig=1, bias=bias, act=K.ACT_GELU, pre_out=h,
                                        ldp=ldh, pre_deriv=True)
                    if pw:
                        continue
                elif name == "pw2 fwd bias+res":       # out = g W2t^T + b + res
                    g, ldg = padded(M, H, pa)
                    w, ldw = padded(C, H, pw)
                    o, ldo = padded(M, C, 0)
                    r, ldr = padded(M, C, 0)
                    fn = lambda: K.gemm(g, w, o, M, C, H, lda=ldg, ldb=ldw, ldd=ldo, a_kcontig=1, b_kcontig=1, bias=bias_c, residual=r, ldr=ldr)
                elif name == "dgrad-pw2 mul aux":      # dh = (dbr W2) * h
                    w, ldw = padded(H, C, 0)
                    d, ldd = padded(M, H, pa)
                    h, ldh = padded(M, H, pa)
                    fn = lambda: K.gemm(x, w, d, M, H, C, lda=ldx, ldb=ldw, ldd=ldd, a_kcontig=1, b_kcontig=1, act=K.ACT_MUL_AUX, aux=h, ldaux=ldh)
                    if pw:
                        continue
                else:                                   # dy2 = dh W1^T
                    d, ldd = padded(M, H, pa)
                    w, ldw = padded(C, H, pw)
                    o, ldo = padded(M, C, 0)
                    fn = lambda: K.gemm(d, w, o, M, C, H, lda=ldd, ldb=ldw, ldd=ldo, a_kcontig=1, b_kcontig=1)
                line += f" | pad {pa}/{pw}: {timeit(fn):6.1f}"
        print(line, flush=True)
