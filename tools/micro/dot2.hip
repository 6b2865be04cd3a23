// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dot2.hip -o tools/micro/dot2   (the binary is git-ignored)
// Issue rate of v_dot2_f32_bf16 (two bf16 MACs per lane, fp32 accumulator, no unpack) against v_pk_fma_f32 (two fp32 FMAs per lane) and
// v_fma_f32, and of the v_perm_b32 that pairs neighbouring pixels: decides whether the depthwise kernels should multiply pixel PAIRS of one
// channel by dot2 instead of unpacking bf16 to fp32 and using packed FMAs on channel pairs.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float acc[CHAINS];
    f32x2 acc2[CHAINS / 2];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) acc[i] = (float)threadIdx.x + i;
#pragma unroll
    for (int i = 0; i < CHAINS / 2; ++i) acc2[i] = f32x2{(float)threadIdx.x + i, (float)i};
    unsigned xa = __float_as_uint(a), xb = __float_as_uint(b);
    const f32x2 av{a, a * 1.0001f}, bv{b, b * 0.999f};
    unsigned pr[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) pr[i] = xa + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (MODE == 0) {      // dot2, VOP3P form
                    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(xa), "v"(xb));
                } else if (MODE == 1) {      // plain fma
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                } else if (MODE == 2) {      // packed fma: CHAINS / 2 instructions cover the same number of MACs as CHAINS dot2
                    if (i < CHAINS / 2) {
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i]) : "v"(av), "v"(bv));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i]) : "v"(bv), "v"(av));
                    }
                } else if (MODE == 3) {      // dot2 with distinct register operands per chain
                    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(pr[i]), "v"(pr[(i + 1) % CHAINS]));
                } else if (MODE == 4) {      // perm only
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pr[i]) : "v"(pr[(i + 1) % CHAINS]), "v"(xb), "v"(xa));
                } else if (MODE == 5) {      // the depthwise mix: 1 perm per 8 dot2
                    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(pr[i]), "v"(pr[(i + 1) % CHAINS]));
                    if ((i & 7) == 0) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pr[i]) : "v"(pr[(i + 1) % CHAINS]), "v"(xb), "v"(xa));
                } else if (MODE == 6) {      // unpack (shift / and) + packed fma: today's depthwise mix, 2 unpack per 7 pk_fma-pairs
                    if (i < CHAINS / 2) {
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i]) : "v"(av), "v"(bv));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i]) : "v"(bv), "v"(av));
                        if ((i & 3) == 0) {
                            asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(pr[i]) : "v"(pr[i + 1]));
                            asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(pr[i + 1]) : "v"(pr[i]));
                        }
                    }
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += acc[i] + __uint_as_float(pr[i]);
#pragma unroll
    for (int i = 0; i < CHAINS / 2; ++i) s += acc2[i].x + acc2[i].y;
    if (s == 1.2345f) out[0] = s;
}

template <int MODE, int CHAINS> void run(const char* name, int waves_per_simd, double macs_per_inst) {
    float* out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * waves_per_simd, iters = 4000;
    k<MODE, CHAINS><<<grid, 256>>>(out, 10, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, CHAINS><<<grid, 256>>>(out, iters, 0.5f, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double slots = (double)iters * 8 * CHAINS;      // loop slots per wavefront
    // cycles per slot and SIMD at 2.4 GHz: waves_per_simd wavefronts share one SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / (slots * waves_per_simd);
    const double macs = (double)grid * 256 * slots * macs_per_inst;
    printf("%-44s chains %2d waves/SIMD %d: %7.3f ms  %5.2f cycles per slot  %6.1f T MAC/s\n", name, CHAINS, waves_per_simd, ms, cyc, macs / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0, 16>("v_dot2_f32_bf16 (2 MAC)", w, 2);
        run<3, 16>("v_dot2_f32_bf16 vgpr x vgpr", w, 2);
        run<1, 16>("v_fma_f32 (1 MAC)", w, 1);
        run<2, 16>("v_pk_fma_f32 (slot = 1 pk_fma = 2 MAC)", w, 2);
        run<4, 16>("v_perm_b32", w, 0);
        run<5, 16>("8 dot2 + 1 perm", w, 2);
        run<6, 16>("7 pk_fma + 2 unpack (per 8 slots)", w, 2);
    }
    return 0;
}
