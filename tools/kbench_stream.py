#!/usr/bin/env python3
"""Streaming (memory-bound) kernels at the shapes of the Swin-T + FPN head (cfg3): BatchNorm statistics / apply / backward, axpby, replace_nan_or_inf,
bilinear resize -- microseconds per launch and the HBM rate by algorithmic bytes, with torch's copy and add as the yardstick of what the part streams.

  python tools/kbench_stream.py [N H W C ...]      default: 16 128 128 256 and 16 64 64 256
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iseg_amd import kernels as K  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def line(name, us, nbytes):
    print(f"  {name:34s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s")


def shape(N, H, W, C):
    rows = N * H * W
    print(f"[{N},{H},{W},{C}] bf16: {rows * C * 2 / 1e6:.0f} MB per tensor")
    dev = "cuda"
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    B = rows * C * 2
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    line("torch copy_ (r+w)", timeit(lambda: y.copy_(x)), 2 * B)
    line("torch add (2r+w)", timeit(lambda: torch.add(x, dy, out=y)), 3 * B)
    line("axpby (2r+w)", timeit(lambda: K.axpby(x, dy, 1.0, 1.0, out=y)), 3 * B)
    packed = K.bn_stats(x, C, rows, C)
    line("bn_stats (r)", timeit(lambda: K.bn_stats(x, C, rows, C, out=packed)), B)
    mean, rstd = K.bn_finalize_apply(packed, x, C, gamma, beta, y, C, rows, C, 1e-3, 0.99, mm, mv, True)
    line("bn_finalize_apply relu (r+w)", timeit(lambda: K.bn_finalize_apply(packed, x, C, gamma, beta, y, C, rows, C, 1e-3, 0.99, mm, mv, True)), 2 * B)
    line("bn_apply_fwd relu (r+w)", timeit(lambda: K.bn_apply_fwd(x, C, mean, rstd, gamma, beta, y, C, rows, C, True)), 2 * B)
    sums = K.bn_bwd_reduce(dy, C, x, C, y, C, mean, rstd, rows, C, True)
    line("bn_bwd_reduce relu (3r)", timeit(lambda: K.bn_bwd_reduce(dy, C, x, C, y, C, mean, rstd, rows, C, True, out=sums)), 3 * B)
    dx = torch.empty_like(x)
    line("bn_bwd_apply relu (3r+w)", timeit(lambda: K.bn_bwd_apply(dy, C, x, C, y, C, mean, rstd, gamma, sums, 1.0 / rows, dx, C, rows, C, True)), 4 * B)
    line("replace_nan_or_inf (2r+w)", timeit(lambda: K.replace_nan_or_inf(x)), 3 * B)
    line("replace_nan_or_inf_bwd (2r+w)", timeit(lambda: K.replace_nan_or_inf_bwd(x, dy)), 3 * B)
    if H % 2 == 0:
        xs = x.reshape(N, H, W, C)[:, ::2, ::2].contiguous()
        line("resize_bilinear x2 (r/4+w)", timeit(lambda: K.resize_bilinear(xs, H, W)), B + B // 4)
        d4 = dy.reshape(N, H, W, C)
        line("resize_bilinear_bwd x2 (r+w/4)", timeit(lambda: K.resize_bilinear_bwd(d4, H // 2, W // 2, torch.bfloat16)), B + B // 4)


def cold(N, H, W, C, R=10):
    """the same launches on R rotating copies of every tensor (working set far beyond the 256 MB Infinity Cache): what a training step sees"""
    rows = N * H * W
    dev = "cuda"
    B = rows * C * 2
    xs = [torch.randn(rows, C, device=dev).to(torch.bfloat16) for _ in range(R)]
    ds = [torch.randn(rows, C, device=dev).to(torch.bfloat16) for _ in range(R)]
    ys = [torch.empty_like(xs[0]) for _ in range(R)]
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    k = [0]

    def rot(fn):
        def run():
            i = k[0] = (k[0] + 1) % R
            fn(xs[i], ds[i], ys[i])
        return run
    print(f"[{N},{H},{W},{C}] bf16, {R} rotating copies ({3 * R * B / 1e9:.1f} GB working set)")
    line("torch copy_ (r+w)", timeit(rot(lambda x, d, y: y.copy_(x))), 2 * B)
    line("torch add (2r+w)", timeit(rot(lambda x, d, y: torch.add(x, d, out=y))), 3 * B)
    line("axpby (2r+w)", timeit(rot(lambda x, d, y: K.axpby(x, d, 1.0, 1.0, out=y))), 3 * B)
    packed = K.bn_stats(xs[0], C, rows, C)
    line("bn_stats (r)", timeit(rot(lambda x, d, y: K.bn_stats(x, C, rows, C, out=packed))), B)
    mean, rstd = K.bn_finalize_apply(packed, xs[0], C, gamma, beta, ys[0], C, rows, C, 1e-3, 0.99, mm, mv, True)
    line("bn_finalize_apply relu (r+w)", timeit(rot(lambda x, d, y: K.bn_finalize_apply(packed, x, C, gamma, beta, y, C, rows, C, 1e-3, 0.99, mm, mv, True))), 2 * B)
    sums = K.bn_bwd_reduce(ds[0], C, xs[0], C, ys[0], C, mean, rstd, rows, C, True)
    line("bn_bwd_reduce relu (3r)", timeit(rot(lambda x, d, y: K.bn_bwd_reduce(d, C, x, C, y, C, mean, rstd, rows, C, True, out=sums))), 3 * B)
    dx = torch.empty_like(xs[0])
    line("bn_bwd_apply relu (3r+w)", timeit(rot(lambda x, d, y: K.bn_bwd_apply(d, C, x, C, y, C, mean, rstd, gamma, sums, 1.0 / rows, dx, C, rows, C, True))), 4 * B)


def fpn(N=16, H=128, W=128, C=768):
    """one FPN level at the Swin-T + FPN stride-4 shape: the fused kernel against its parts"""
    dev = "cuda"
    z = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    xc = torch.randn(N, H // 2, W // 2, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    B = z.numel() * 2
    rows = N * H * W
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    packed = K.bn_stats(z.reshape(rows, C), C, rows, C)
    mean, rstd = K.bn_finalize(packed, C, 1e-3, 0.9, None, None)
    print(f"[{N},{H},{W},{C}] bf16: {B / 1e6:.0f} MB per tensor")
    line("bn_relu_upsample_add (r + r/4 + w)", timeit(lambda: K.bn_relu_upsample_add(z, mean, rstd, g, b, xc)), 2 * B + B // 4)
    line("resize_bilinear_bwd x2 (r + w/4)", timeit(lambda: K.resize_bilinear_bwd(dy, H // 2, W // 2, torch.bfloat16)), B + B // 4)
    z2, d2 = z.reshape(rows, C), dy.reshape(rows, C)
    sums = K.bn_bwd_reduce_remask(d2, C, z2, C, mean, rstd, g, b, rows, C)
    line("bn_bwd_reduce_remask (2r)", timeit(lambda: K.bn_bwd_reduce_remask(d2, C, z2, C, mean, rstd, g, b, rows, C, out=sums)), 2 * B)
    dz = torch.empty_like(z2)
    line("bn_bwd_apply_remask (2r + w)", timeit(lambda: K.bn_bwd_apply_remask(d2, C, z2, C, mean, rstd, g, b, sums, 1.0 / rows, dz, C, rows, C)), 3 * B)
    line("bn_stats (r)", timeit(lambda: K.bn_stats(z2, C, rows, C, out=packed)), B)


def main():
    if sys.argv[1:2] == ["fpn"]:
        return fpn(*[int(v) for v in sys.argv[2:]])
    if sys.argv[1:2] == ["cold"]:
        a = [int(v) for v in sys.argv[2:]] or [16, 128, 128, 256]
        return cold(*a)
    a = [int(v) for v in sys.argv[1:]]
    shapes = [a[i:i + 4] for i in range(0, len(a), 4)] or [[16, 128, 128, 256], [16, 64, 64, 256], [16, 128, 128, 96]]
    for s in shapes:
        shape(*s)


if __name__ == "__main__":
    main()
