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
                } else if (MODE == 10) asm volatile("v_pk_fma_f16 %0, %1, %0, %2" : "+v"(v) : "v"(a), "v"(b));
                else if (MODE == 11) asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(v) : "v"(a));
                else if (MODE == 12) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(v) : "v"(a));
                else if (MODE == 13) asm volatile("v_pk_min_f16 %0, %1, %0" : "+v"(v) : "v"(a));
                else if (MODE == 14) asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(v) : "v"(a));
                else if (MODE == 15) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v) : "v"(a));
                else if (MODE == 16) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]" : "+v"(v) : "v"(a), "v"(b));
                else if (MODE == 17) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v));
                else if (MODE == 18) asm volatile("v_exp_f16 %0, %0" : "+v"(v));
                else if (MODE == 19) asm volatile("v_rcp_f16 %0, %0" : "+v"(v));
                else if (MODE == 20) asm volatile("v_fma_f16 %0, %1, %0, %2" : "+v"(v) : "v"(a), "v"(b));
                else if (MODE == 21) asm volatile("v_fma_mixlo_f16 %0, %1, %0, %2 op_sel_hi:[0,0,0]" : "+v"(v) : "v"(a), "v"(b));
                else if (MODE == 22) asm volatile("v_pk_fma_f16 %0, %1, %0, %2 clamp" : "+v"(v) : "v"(a), "v"(b));
                else if (MODE == 23) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(*reinterpret_cast<double*>(&acc[i & ~1])) : "v"(*reinterpret_cast<double*>(&acc[(i & ~1) ^ 2])));
                else if (MODE == 24) {      // packed-f16 gelu of TWO elements (slot = 2 evaluations): pkrtz + clamp(2) + t^2 + 4 fma + fma + mul
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    const h2 x = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(v, v + a));
                    const h2 lim = {(_Float16)4.f, (_Float16)4.f};
                    h2 t = __builtin_elementwise_min(__builtin_elementwise_max(x, -lim), lim);
                    const h2 t2 = t * t;
                    h2 q = (h2){(_Float16)1e-4f, (_Float16)1e-4f} * t2 + (h2){(_Float16)-3e-3f, (_Float16)-3e-3f};
                    q = q * t2 + (h2){(_Float16)2e-2f, (_Float16)2e-2f};
                    q = q * t2 + (h2){(_Float16)-6e-2f, (_Float16)-6e-2f};
                    q = q * t2 + (h2){(_Float16)0.39f, (_Float16)0.39f};
                    const h2 phi = t * q + (h2){(_Float16)0.5f, (_Float16)0.5f};
                    const h2 g = x * phi;
                    v = __builtin_bit_cast(float, g);
                } else if (MODE == 25) {      // packed-f16 gelu' of TWO elements times an fp32 factor each (v_fma_mix_f32), packed to bf16
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    const h2 x = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(v, v + a));
                    const h2 lim = {(_Float16)4.f, (_Float16)4.f};
                    h2 t = __builtin_elementwise_min(__builtin_elementwise_max(x, -lim), lim);
                    const h2 t2 = t * t;
                    h2 q = (h2){(_Float16)1e-4f, (_Float16)1e-4f} * t2 + (h2){(_Float16)-3e-3f, (_Float16)-3e-3f};
                    q = q * t2 + (h2){(_Float16)2e-2f, (_Float16)2e-2f};
                    q = q * t2 + (h2){(_Float16)-6e-2f, (_Float16)-6e-2f};
                    q = q * t2 + (h2){(_Float16)-0.2f, (_Float16)-0.2f};
                    q = q * t2 + (h2){(_Float16)0.79f, (_Float16)0.79f};
                    const h2 d = t * q + (h2){(_Float16)0.5f, (_Float16)0.5f};
                    const float d0 = (float)d[0] * a, d1 = (float)d[1] * b;
                    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                    b2 o;
                    o[0] = (__bf16)d0;
                    o[1] = (__bf16)d1;
                    v = __builtin_bit_cast(float, o);
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
        run<10, 16>("v_pk_fma_f16 (2 elements)", w);
        run<22, 16>("v_pk_fma_f16 clamp", w);
        run<11, 16>("v_pk_mul_f16", w);
        run<12, 16>("v_pk_add_f16", w);
        run<13, 16>("v_pk_min_f16", w);
        run<14, 16>("v_pk_max_f16", w);
        run<15, 16>("v_cvt_pkrtz_f16_f32", w);
        run<16, 16>("v_fma_mix_f32 (f16 x f32 + f32)", w);
        run<21, 16>("v_fma_mixlo_f16", w);
        run<17, 16>("v_cvt_f32_f16", w);
        run<18, 16>("v_exp_f16", w);
        run<19, 16>("v_rcp_f16", w);
        run<20, 16>("v_fma_f16", w);
        run<23, 16>("v_pk_mul_f32", w);
        run<24, 16>("gelu packed f16 poly (slot = 2 values)", w);
        run<25, 16>("gelu' packed f16 x f32 -> bf16 (slot = 2)", w);
    }
    return 0;
}
