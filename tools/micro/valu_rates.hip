// build: hipcc --offload-arch=gfx950 -O3 -w tools/micro/valu_rates.hip -o tools/micro/valu_rates   (the binary is git-ignored)
// Issue cost (cycles per wave-instruction and SIMD) of the instructions a GELU evaluation is made of: plain fp32 (mul / fma / min / med3),
// the two transcendentals of the sigmoid form (v_exp_f32, v_rcp_f32), the bf16 pack, and two whole GELU forms (sigmoid with exp + rcp,
// clamped odd polynomial without transcendentals) -- 16 independent chains per lane, 1 / 2 / 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float acc[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) acc[i] = 0.001f * (float)threadIdx.x + i * 0.37f - 2.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                float& v = acc[i];
                if (MODE == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v) : "v"(a));
                else if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(v) : "v"(a), "v"(b));
                else if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
                else if (MODE == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(v));
                else if (MODE == 4) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
                else if (MODE == 5) asm volatile("v_min_f32 %0, %1, %0" : "+v"(v) : "v"(a));
                else if (MODE == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v) : "v"(a));
                else if (MODE == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
                else if (MODE == 8) {      // sigmoid-form gelu: 7 plain + exp + rcp
                    const float x = v;
                    const float x2 = fminf(x * x, 64.f);
                    const float p = fmaf(fmaf(0.001f, x2, -0.1f), x2, -2.3f);
                    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
                    v = x * s + a;
                } else if (MODE == 9) {      // clamped odd polynomial, degree 6 in x^2: 1 med3 + 1 mul + 6 fma + 1 fma + 1 mul
                    const float x = v;
                    const float xc = __builtin_amdgcn_fmed3f(x, -b, b);
                    const float t = xc * xc;
                    float p = fmaf(1e-7f, t, -1e-5f);
                    p = fmaf(p, t, 3e-4f);
                    p = fmaf(p, t, -4e-3f);
                    p = fmaf(p, t, 3e-2f);
                    p = fmaf(p, t, -0.13f);
                    p = fmaf(p, t, 0.39f);
                    const float phi = fmaf(xc, p, 0.5f);
                    v = x * phi + a;
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += acc[i];
    if (s == 1.2345f) out[0] = s;
}

template <int MODE, int CHAINS> void run(const char* name, int waves_per_simd) {
    float* out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * waves_per_simd, iters = 4000;
    k<MODE, CHAINS><<<grid, 256>>>(out, 10, 0.5f, 3.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, CHAINS><<<grid, 256>>>(out, iters, 0.5f, 3.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double slots = (double)iters * 4 * CHAINS;
    const double cyc = ms * 1e-3 * 2.4e9 / (slots * waves_per_simd);
    printf("%-34s waves/SIMD %d: %7.3f ms  %6.2f cycles per slot and SIMD\n", name, waves_per_simd, ms, cyc);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0, 16>("v_mul_f32", w);
        run<1, 16>("v_fma_f32", w);
        run<7, 16>("v_fmac_f32", w);
        run<2, 16>("v_exp_f32", w);
        run<3, 16>("v_rcp_f32", w);
        run<4, 16>("v_med3_f32", w);
        run<5, 16>("v_min_f32", w);
        run<6, 16>("v_cvt_pk_bf16_f32", w);
        run<8, 16>("gelu sigmoid form (whole)", w);
        run<9, 16>("gelu clamped polynomial (whole)", w);
    }
    return 0;
}
