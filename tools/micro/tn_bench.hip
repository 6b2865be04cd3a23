// Timing harness for gemm_bf16_dma_tn_kernel and its ablation builds:
//   hipcc --offload-arch=gfx950 -O3 -w -Iinclude -Iiseg_amd/csrc [-DISEG_TN_ABL_NOWAIT] [-DISEG_TN_ABL_NODMA] [-DISEG_TN_ABL_NOMFMA] tools/micro/tn_bench.hip -o tools/micro/tnb_<name>
//   ./tools/micro/tnb_<name> [M N K splits]        (defaults: the stage-2 pwconv2 weight gradient 1536 384 16384 13)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
extern "C" void iseg_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
int iseg_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }
#include "../../iseg_amd/csrc/gemm_dma_tn.h"
namespace iseg_mm {
int dma_mode() { return 1; }
int dma_tn_mode() { return 1; }
int long_k_tile() { return 128; }
int tile_waves() { return 8; }
}

int main(int argc, char** argv) {
    const int64_t M = argc > 4 ? atoll(argv[1]) : 1536, N = argc > 4 ? atoll(argv[2]) : 384, K = argc > 4 ? atoll(argv[3]) : 16384;
    const int want = argc > 4 ? atoi(argv[4]) : 13;
    const int64_t kps = ((K + want - 1) / want + 127) / 128 * 128;
    const int nsplit = (int)((K + kps - 1) / kps);
    void *A, *B; float* slabs;
    hipMalloc(&A, K * M * 2); hipMalloc(&B, K * N * 2); hipMalloc(&slabs, (size_t)nsplit * (M + 1) * N * 4);
    hipMemset(A, 0x3c, K * M * 2); hipMemset(B, 0x3b, K * N * 2);
    iseg_gemm_args g = {};
    g.A = A; g.B = B; g.lda = M; g.ldb = N; g.M = M; g.N = N; g.K = K; g.in_dtype = ISEG_BF16; g.batch = 1; g.colsum_out = slabs;
    const bool wide = (M + 127) / 128 * ((N + 255) / 256) < (M + 255) / 256 * ((N + 127) / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        const int n = 20;
        for (int i = 0; i < n; ++i) {
            if (wide) iseg_mm::launch_dma_tn<2, 4>(&g, nsplit, kps, slabs, 0);
            else iseg_mm::launch_dma_tn<4, 2>(&g, nsplit, kps, slabs, 0);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("M=%lld N=%lld K=%lld splits=%d (%lld rows each) tile %s: %.1f us per launch, %.0f TFLOP/s\n", (long long)M, (long long)N, (long long)K, nsplit,
                        (long long)kps, wide ? "128x256" : "256x128", ms * 1e3 / n, 2.0 * M * N * K / (ms * 1e-3 / n) / 1e12);
    }
    if (hipGetLastError() != hipSuccess) return 1;
    return 0;
}
