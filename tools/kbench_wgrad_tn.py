#!/usr/bin/env python3
"""Weight-gradient GEMM shapes of the flagship (dW = X^T dY, split over the pixels, bias gradient on the ones-row), timed with the reduction of the
slabs and without, on the register-staged kernel (ISEG_GEMM_DMA_TN=0) and on the LDS-DMA kernel of csrc/gemm_dma_tn.h (default).  The knob is read
at first use, so each setting runs in its own process:
  python tools/kbench_wgrad_tn.py            # both settings, all shapes
  python tools/kbench_wgrad_tn.py one        # this process's setting only"""
import os
import subprocess
import sys

SHAPES = [(16384, 1536, 384), (16384, 384, 1536), (4096, 3072, 768), (4096, 768, 3072), (65536, 768, 192), (65536, 192, 768)]
if os.environ.get("KBENCH_NARROW"):      # the narrow stages of Swin-T / InternImage-B (M or N below 128)
    SHAPES = [(262144, 96, 288), (262144, 96, 96), (262144, 96, 384), (262144, 384, 96), (65536, 192, 576), (131072, 112, 336), (131072, 112, 448), (32768, 224, 672), (8192, 448, 1344)]


def one():
    import ctypes as C

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from iseg_amd import _hip
    from iseg_amd import kernels as K

    for rows, Cc, N in SHAPES:
        torch.manual_seed(0)
        x = torch.randn(rows, Cc, device="cuda").bfloat16()
        dy = torch.randn(rows, N, device="cuda").bfloat16()
        dW = torch.zeros(Cc, N, device="cuda")
        db = torch.zeros(N, device="cuda")

        def run():
            K.gemm(x, dy, dW, Cc, N, rows, lda=Cc, ldb=N, ldd=N, a_kcontig=0, b_kcontig=0, accumulate=False, colsum_out=db, colsum_accumulate=False)

        def run_slabs():
            K.dense_wgrad_slabs(x, dy)

        run()
        ref = x.float().t() @ dy.float()
        err = (dW - ref).abs().max().item() / ref.abs().max().item()
        errb = (db - dy.float().sum(0)).abs().max().item() / dy.float().sum(0).abs().max().item()
        g = _hip.GemmArgs()
        g.A, g.lda, g.B, g.ldb, g.D, g.ldd = x.data_ptr(), Cc, dy.data_ptr(), N, dW.data_ptr(), N
        g.M, g.N, g.K, g.in_dtype, g.batch, g.batch_inner, g.colsum_out = Cc, N, rows, 1, 1, 1, db.data_ptr()
        L = _hip.lib()
        res = []
        for fn in (run, run_slabs):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 50)
        tf = 2.0 * rows * Cc * N / (res[1] * 1e-6) / 1e12
        print(f"DMA_TN={os.environ.get('ISEG_GEMM_DMA_TN', '1')} rows {rows:6d} M {Cc:5d} N {N:5d}: variant {int(L.iseg_gemm_variant(C.byref(g)))} "
              f"slabs {int(L.iseg_gemm_slabs(C.byref(g))):3d}  with reduce {res[0]:7.1f} us  product alone {res[1]:7.1f} us ({tf:6.1f} TFLOP/s)  "
              f"rel err dW {err:.1e} db {errb:.1e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for v in ("0", "1"):
            env = dict(os.environ, ISEG_GEMM_DMA_TN=v)
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, check=False)
