// build: hipcc --offload-arch=gfx950 -O3 tools/micro/pkfma.hip -o tools/micro/pkfma   (the binary is git-ignored)
// Issue rate of v_pk_fma_f32 against v_fma_f32 on one SIMD (N independent accumulator chains per lane, W waves per SIMD): decides whether
// the depthwise kernels' packed FMAs run at twice the plain rate (the 157 TF vector peak counts them so) or at the same rate.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x2 acc[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) acc[i] = f32x2{(float)threadIdx.x + i, (float)i};
    const f32x2 av{a, a * 1.0001f}, bv{b, b * 0.999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (MODE == 0) {      // packed: one instruction, two FMAs per lane
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
                } else if (MODE == 1) {      // two plain FMAs
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(av.x), "v"(bv.x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(av.y), "v"(bv.y));
                } else {      // packed with distinct multiplicand registers per chain (the depthwise pattern: x and w both vary)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(acc[(i + 1) % CHAINS]), "v"(bv));
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += acc[i].x + acc[i].y;
    if (s == 1.2345f) out[0] = s;
}

template <int MODE, int CHAINS> void run(const char* name, int waves_per_simd) {
    float* out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * waves_per_simd, iters = 4000;      // 256 threads = 4 waves = one per SIMD; `waves_per_simd` workgroups per CU
    k<MODE, CHAINS><<<grid, 256>>>(out, 10, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, CHAINS><<<grid, 256>>>(out, iters, 0.5f, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)grid * 256 * iters * 8 * CHAINS * 2;
    printf("%-28s chains %2d waves/SIMD %d: %7.3f ms  %6.1f TFLOP/s\n", name, CHAINS, waves_per_simd, ms, 2 * fma / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w : {1, 2}) {
        run<0, 8>("v_pk_fma_f32", w);
        run<1, 8>("2 x v_fma_f32", w);
        run<2, 8>("v_pk_fma_f32 (vgpr x vgpr)", w);
        run<0, 16>("v_pk_fma_f32", w);
        run<1, 16>("2 x v_fma_f32", w);
    }
    return 0;
}
