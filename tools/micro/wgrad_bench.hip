// Timing harness for convnext_mlp_wgrad_kernel variants (compile with -DWG_KNOB=<bits>): hipcc --offload-arch=gfx950 -O3 -Iinclude -Iiseg_amd/csrc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
extern "C" void iseg_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
int iseg_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }
float* iseg_deferred_partials(size_t, const float*, const float*, int, hipStream_t) { return nullptr; }
void iseg_deferred_push(const float*, int, int64_t, int64_t, float*, float*, int64_t, float, hipStream_t) {}
#include "../../iseg_amd/csrc/mlp_wgrad.hip"

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 96;
    const int S = C == 96 ? 128 : 64;
    const int64_t M = 16LL * S * S;
    const int HID = 4 * C;
    void *y, *d, *bw; float *b1, *W2, *b2, *gamma, *g, *rs; void* ws;
    hipMalloc(&y, M * C * 2); hipMalloc(&d, M * C * 2); hipMalloc(&bw, 3 * 4 * C * C * 2);
    hipMalloc(&b1, HID * 4); hipMalloc(&W2, HID * C * 4); hipMalloc(&b2, C * 4); hipMalloc(&gamma, C * 4); hipMalloc(&rs, 64);
    hipMalloc(&g, (2 * HID * C + HID + 2 * C) * 4);
    const size_t wsb = iseg_convnext_mlp_wgrad_workspace_bytes(M, C);
    hipMalloc(&ws, wsb);
    hipMemset(y, 0x3c, M * C * 2); hipMemset(d, 0x3b, M * C * 2); hipMemset(bw, 0x3a, 3 * 4 * C * C * 2);
    hipMemset(b1, 0, HID * 4); hipMemset(W2, 0, HID * C * 4); hipMemset(b2, 0, C * 4); hipMemset(gamma, 0, C * 4); hipMemset(g, 0, (2 * HID * C + HID + 2 * C) * 4);
    float one[16]; for (int i = 0; i < 16; ++i) one[i] = 1.f; hipMemcpy(rs, one, 64, hipMemcpyHostToDevice);
    float *dW1 = g, *db1 = dW1 + HID * C, *dW2 = db1 + HID, *db2 = dW2 + HID * C, *dg = db2 + C;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        const int n = 10;
        for (int i = 0; i < n; ++i)
            if (iseg_convnext_mlp_wgrad(y, nullptr, nullptr, nullptr, nullptr, d, rs, (int64_t)S * S, bw, b1, W2, b2, gamma, dW1, db1, dW2, db2, dg, M, C, 1, ws, wsb, 0)) return 1;
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("C=%d knob=%d: %.1f us per call (kernel + finish)\n", C,
#ifdef WG_KNOB
                        WG_KNOB,
#else
                        0,
#endif
                        ms * 1e3 / n);
    }
    return 0;
}
