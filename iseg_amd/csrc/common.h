// Shared device/host helpers for the iseg_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define ISEG_OK 0
#define ISEG_ERR_ARG (-1)
#define ISEG_ERR_HIP (-2)
#define ISEG_ERR_UNSUPPORTED (-3)
#define ISEG_ERR_WORKSPACE (-4)

#define ISEG_F32 0
#define ISEG_BF16 1

typedef __bf16 bf16_t;

// ---- error plumbing (api.cpp) ----------------------------------------------------------
extern "C" void iseg_set_error(const char* fmt, ...);
int iseg_check_launch(const char* what);

#define ISEG_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            iseg_set_error(__VA_ARGS__);        \
            return ISEG_ERR_ARG;                \
        }                                       \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t dtype_size(int dtype) { return dtype == ISEG_BF16 ? 2 : 4; }

// ---- device helpers ---------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;      // one dword of a bf16 row: operand of v_dot2_f32_bf16

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <class T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte vector access: VecN<T>::N elements per 16 B.
template <class T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ __forceinline__ static void load(const float* p, float* out) {
        float4 v = *reinterpret_cast<const float4*>(p);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    }
    __device__ __forceinline__ static void store(float* p, const float* in) {
        *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
    }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    __device__ __forceinline__ static void load(const bf16_t* p, float* out) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
    }
    __device__ __forceinline__ static void store(bf16_t* p, const float* in) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)in[i];
        *reinterpret_cast<bf16x8*>(p) = v;
    }
};

// 8 consecutive elements (used where bf16 and f32 kernels share a "8 channels per lane" shape).
template <class T> __device__ __forceinline__ void load8(const T* p, float* out);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* out) {
    Vec16<float>::load(p, out);
    Vec16<float>::load(p + 4, out + 4);
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* out) { Vec16<bf16_t>::load(p, out); }
template <class T> __device__ __forceinline__ void store8(T* p, const float* in);
template <> __device__ __forceinline__ void store8<float>(float* p, const float* in) {
    Vec16<float>::store(p, in);
    Vec16<float>::store(p + 4, in + 4);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* in) { Vec16<bf16_t>::store(p, in); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over aligned groups of `width` lanes (width power of two <= 64)
// Sum over aligned groups of `width` lanes (a power of two <= 64), result in every lane of the group.  Cross-lane steps are DPP / permlane
// VALU operations (quad_perm, row_half_mirror, row_mirror inside a 16-lane row; v_permlane16_swap / v_permlane32_swap between rows), not
// ds_bpermute round trips through the LDS crossbar (__shfl_xor): a 64-lane sum is 6 short VALU steps instead of 6 dependent LDS operations.
#define ISEG_DPP_STEP(V, CTRL) ((V) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (V)), (CTRL), 0xf, 0xf, true)))
__device__ __forceinline__ float group_sum(float v, int width) {
    if (width >= 2) v = ISEG_DPP_STEP(v, 0xB1);       // quad_perm [1,0,3,2]
    if (width >= 4) v = ISEG_DPP_STEP(v, 0x4E);       // quad_perm [2,3,0,1]
    if (width >= 8) v = ISEG_DPP_STEP(v, 0x141);      // row_half_mirror: the other quad of each 8
    if (width >= 16) v = ISEG_DPP_STEP(v, 0x140);     // row_mirror: the other half of each 16
    if (width >= 32) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // [r0 r0 r2 r2], [r1 r1 r3 r3]
        v = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
    if (width >= 64) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // [lo lo], [hi hi]
        v = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
    return v;
}

// Deferred parameter-gradient reductions (api.hip).  Between iseg_deferred_begin() and iseg_deferred_end() a two-stage reduction whose
// output lies inside the registered gradient buffer (and accumulates into it) is not launched: its partial sums go to the caller's arena and
// a descriptor is queued; iseg_deferred_flush() runs ONE launch over all queued descriptors.  The ~70 five-microsecond reduce launches of a
// training step (LayerNorm / depthwise / bias gradients) cost more at their kernel boundaries than in their work.
//   iseg_deferred_partials(): arena slice for `bytes` of partials if this reduction may be deferred, else nullptr (launch it right away)
float* iseg_deferred_partials(size_t bytes, const float* out0, const float* out1, int accumulate, hipStream_t stream);
void iseg_deferred_push(const float* partials, int P, int64_t pstride, int64_t n, float* out0, float* out1, int64_t n0, float scale,
                        hipStream_t stream);

// Second stage of every two-stage reduction (LayerNorm / BatchNorm / colsum / depthwise-wgrad parameter gradients):
//   out[b][j] (+)= scale * sum_{p<P} partials[b*bstride_in + p*pstride + j],  j < n,
// written to out0 for j < n0 and to out1[j-n0] beyond (out1 may be null).  A block owns 16 columns; 16 row-lanes stride
// over p and are combined through LDS in a fixed order, so the result is deterministic and the pass takes P/64 dependent
// steps instead of P.  Template parameter only makes the symbol TU-local-safe.
template <int TAG>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partials, int P, int64_t pstride,
                                                          int64_t bstride_in, int64_t n, float* __restrict__ out0,
                                                          float* __restrict__ out1, int64_t n0, int64_t bstride_out, float scale,
                                                          int accumulate) {
    // 16 columns x 16 row-lanes per workgroup: P/16 dependent steps, 64-B coalesced segments
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 16 + tx;
    const float* src = partials + (int64_t)blockIdx.y * bstride_in;
    float s = 0.f;
    if (j < n) {
        int p = ty;
        for (; p + 48 < P; p += 64) {  // four independent loads in flight
            const float a = src[(int64_t)p * pstride + j], b = src[(int64_t)(p + 16) * pstride + j];
            const float c = src[(int64_t)(p + 32) * pstride + j], d = src[(int64_t)(p + 48) * pstride + j];
            s += (a + b) + (c + d);
        }
        for (; p < P; p += 16) s += src[(int64_t)p * pstride + j];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && j < n) {
        float t = red[0][tx];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][tx];
        t *= scale;
        float* dst = j < n0 ? out0 + (int64_t)blockIdx.y * bstride_out + j : (out1 ? out1 + (int64_t)blockIdx.y * bstride_out + (j - n0) : nullptr);
        if (dst) {
            if (accumulate) t += *dst;
            *dst = t;
        }
    }
}

// Wide variant for long partial rows (split-K slabs of the big weight gradients, chunk records: n ~ 10^5..10^6, P <= 128): a lane owns
// four consecutive columns (16-B loads, 1 KiB per wave per row) and walks the P rows in order -- same fixed summation order
// for every element, no LDS.  Requires pstride % 4 == 0, n % 4 == 0, n0 % 4 == 0 and 16-B aligned bases (checked on the host).
template <int TAG>
__global__ __launch_bounds__(256) void reduce_rows_wide_kernel(const float* __restrict__ partials, int P, int64_t pstride,
                                                               int64_t n, float* __restrict__ out0, float* __restrict__ out1,
                                                               int64_t n0, float scale, int accumulate) {
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= n) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = 0;
    for (; p + 3 < P; p += 4) {
        const float4 a = *reinterpret_cast<const float4*>(partials + (int64_t)p * pstride + j);
        const float4 b = *reinterpret_cast<const float4*>(partials + (int64_t)(p + 1) * pstride + j);
        const float4 c = *reinterpret_cast<const float4*>(partials + (int64_t)(p + 2) * pstride + j);
        const float4 d = *reinterpret_cast<const float4*>(partials + (int64_t)(p + 3) * pstride + j);
        s.x += (a.x + b.x) + (c.x + d.x);
        s.y += (a.y + b.y) + (c.y + d.y);
        s.z += (a.z + b.z) + (c.z + d.z);
        s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; p < P; ++p) {
        const float4 a = *reinterpret_cast<const float4*>(partials + (int64_t)p * pstride + j);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
    float* dst = j < n0 ? out0 + j : (out1 ? out1 + (j - n0) : nullptr);
    if (!dst) return;
    if (accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(dst);
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *reinterpret_cast<float4*>(dst) = s;
}

static inline void launch_reduce_rows(const float* partials, int P, int64_t pstride, int64_t bstride_in, int batch, int64_t n,
                                      float* out0, float* out1, int64_t n0, int64_t bstride_out, float scale, int accumulate,
                                      hipStream_t stream) {
    const bool aligned = (((uintptr_t)partials | (uintptr_t)out0 | (uintptr_t)out1) & 15) == 0;
    // (up to 128 partial rows: the chunk records of the fused-stage weight gradients are 40 / 80 rows of 0.3 M floats -- the 16 x 16 kernel below
    // reads them in 64-byte segments at 2.8 TB/s; flagship 8.91 -> 8.87 ms.  ISEG_REDUCE_WIDE_MAXP overrides.)
    static const int wide_max_p = [] { const char* e = getenv("ISEG_REDUCE_WIDE_MAXP"); return e ? atoi(e) : 128; }();
    if (batch == 1 && P <= wide_max_p && n >= 32768 && n % 4 == 0 && n0 % 4 == 0 && pstride % 4 == 0 && aligned) {
        hipLaunchKernelGGL((reduce_rows_wide_kernel<0>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, partials, P,
                           pstride, n, out0, out1, n0, scale, accumulate);
        return;
    }
    hipLaunchKernelGGL((reduce_rows_kernel<0>), dim3((unsigned)((n + 15) / 16), batch), dim3(256), 0, stream, partials, P, pstride,
                       bstride_in, n, out0, out1, n0, bstride_out, scale, accumulate);
}

// Fast Phi(x) = 0.5(1+erf(x/sqrt2)) for bf16-storage kernels: Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the
// bf16 output rounding of 2^-9), one v_exp + one v_rcp instead of libm erff's ~50 VALU instructions -- the pwconv1 GEMM
// of ConvNeXt evaluates 100 M GELUs per launch at stage 0 and was VALU-bound on erff.  `e` returns exp(-x^2/2).
__device__ __forceinline__ float norm_cdf_fast(float x, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    e = __expf(-0.5f * x * x);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float h = 0.5f * p * t * e;  // 0.5 * erfc(|x|/sqrt2)
    return x >= 0.f ? 1.0f - h : h;
}
__device__ __forceinline__ float gelu_fast(float x) {
    float e;
    return x * norm_cdf_fast(x, e);
}
__device__ __forceinline__ float gelu_fast_grad(float x) {
    float e;
    const float cdf = norm_cdf_fast(x, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}

// gelu(x) and gelu'(x) from one Phi / exp evaluation (forward epilogue that saves the derivative for the backward pass)
__device__ __forceinline__ void gelu_fast_both(float x, float& y, float& dy) {
    float e;
    const float cdf = norm_cdf_fast(x, e);
    y = x * cdf;
    dy = fmaf(x * 0.39894228040143267794f, e, cdf);
}

// Cheapest GELU that is still far below bf16 rounding (used by the fused ConvNeXt MLP kernels, whose hidden tile never leaves the
// CU and is rounded to bf16 as the next MFMA operand): Phi(x) ~ sigmoid(x * (a0 + a1 x^2 + a2 x^4)), coefficients from a minimax
// fit of x * Phi(x) on [-8, 8] (tools/fit_gelu_sigmoid.py): |gelu error| <= 2.6e-5, |gelu' error| <= 1.1e-4 (bf16 half-ulp at 1 is
// 2e-3).  One v_exp_f32 + one v_rcp_f32 + 7 plain VALU per element instead of the ~27 issue slots of the erfc form above.
// The polynomial's x^4 coefficient is negative, so x^2 is clamped at 64 (|x| > 8: sigmoid is saturated either way).
#define ISEG_GELU_SIG_A0 1.5950157270240881f
#define ISEG_GELU_SIG_A1 0.07401132728640801f
#define ISEG_GELU_SIG_A2 (-0.0007030389408329068f)
// gelu(x) AND gelu'(x) of the bf16 kernels that need both (weight-gradient recompute, the forward GEMM epilogue that saves the derivative): the
// two-coefficient sigmoid fit -- no clamp (its polynomial is monotone), 9 full-rate VALU + v_exp_f32 + v_rcp_f32 for the pair;
// |gelu error| <= 2.7e-4, |gelu' error| <= 8.7e-4 (bf16 half-ulp at 1: 2e-3; the reference evaluates GELU in bf16 arithmetic under
// mixed_bfloat16, utils/common.py:32-64 + backbones/convnext.py:53, i.e. with ~4e-3 per operation).  Round 5: measured 39 issue cycles per
// pair against 48 for the three-coefficient fit and 45 for two polynomials (tools/micro/valu_rates.hip, profiles/r05_micro.txt).
// ISEG_GELU_SIG3 (A/B build define) restores the three-coefficient fit (2.6e-5 / 1.1e-4).
#ifndef ISEG_GELU_SIG3
__device__ __forceinline__ float gelu_sig(float x) {
    constexpr float L = -1.4426950408889634f;
    const float p = fmaf(0.06940208738399849f * L, x * x, 1.600313485784997f * L);
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
    return x * s;
}
__device__ __forceinline__ void gelu_sig_both(float x, float& y, float& dy) {
    constexpr float L = -1.4426950408889634f;
    const float x2 = x * x;
    const float p = fmaf(0.06940208738399849f * L, x2, 1.600313485784997f * L);
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
    const float dp = fmaf(3.f * 0.06940208738399849f, x2, 1.600313485784997f);
    y = x * s;
    dy = fmaf(y * dp, 1.0f - s, s);
}
#else
__device__ __forceinline__ float gelu_sig(float x) {
    const float x2 = fminf(x * x, 64.f);
    constexpr float L = -1.4426950408889634f;      // -log2(e): sigmoid(u) = 1 / (1 + exp2(-u * log2 e))
    const float p = fmaf(fmaf(ISEG_GELU_SIG_A2 * L, x2, ISEG_GELU_SIG_A1 * L), x2, ISEG_GELU_SIG_A0 * L);
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
    return x * s;
}
// gelu(x) and d/dx of the SAME approximation: s + x s (1 - s) (a0 + 3 a1 x^2 + 5 a2 x^4)
__device__ __forceinline__ void gelu_sig_both(float x, float& y, float& dy) {
    const float x2 = fminf(x * x, 64.f);
    constexpr float L = -1.4426950408889634f;
    const float p = fmaf(fmaf(ISEG_GELU_SIG_A2 * L, x2, ISEG_GELU_SIG_A1 * L), x2, ISEG_GELU_SIG_A0 * L);
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
    const float dp = fmaf(fmaf(5.f * ISEG_GELU_SIG_A2, x2, 3.f * ISEG_GELU_SIG_A1), x2, ISEG_GELU_SIG_A0);
    y = x * s;
    dy = fmaf(y * dp, 1.0f - s, s);
}
#endif

// Transcendental-free GELU / GELU' for the bf16 kernels that need ONE of the two (round 5; fits: tools/fit_gelu_poly.py).  Every instruction is
// on the vector pipe's full-rate path (v_fma_f32 / v_mul_f32 / v_add_f32: 2.3-2.7 cycles per wave-instruction at two wavefronts per SIMD; v_exp /
// v_rcp cost 8.3, v_min / v_med3 / every v_pk_*_f16 4.4 -- packed f16 buys nothing over this form, profiles/r05_micro.txt):
//     s = clamp01(x / (2 c) + 1/2)      ONE v_fma_f32 with the clamp output modifier
//     w = s - 1/2                       = clamp(x, -c, c) / (2 c)
//     Phi(x)   ~ 1/2 + w q(w^2),  q(1/4) = 1  (c = 3.75, degree 6):  |gelu error| <= 8.9e-5 max(1, |x|); x (1 - 1e-6) / 1e-6 x beyond the clamp
//     gelu'(x) ~ 1/2 + w r(w^2),  r(1/4) = 1  (c = 4,    degree 7):  |gelu' error| <= 5.2e-4
// (Phi - 1/2 and gelu' - 1/2 are odd, so q and r are polynomials in w^2.)  gelu: 11 instructions = 26 cycles against 41 for the sigmoid form;
// gelu' alone: 12 against the 48 of gelu_sig_both.
// clamp01(a * b + 1/2) as ONE v_fma_f32 ... clamp.  With a literal multiplier hipcc emits v_fmamk_f32 (VOP2: no output modifier) + v_max_f32 ...
// clamp (half rate); with the multiplier in a scalar register it must use the VOP3 form and folds the clamp into it.  The constant therefore
// goes through an opaque s_mov.  (NOT inline assembly for the FMA itself: the hazard recogniser does not look inside an asm statement, and an asm
// v_fma that reads an MFMA accumulator is issued without the wait states the matrix pipe needs -- the first version of this read stale values.)
__device__ __forceinline__ float iseg_opaque_scalar(float v) {
    float k;
    asm("s_mov_b32 %0, %1" : "=s"(k) : "s"(v));
    return k;
}
__device__ __forceinline__ float iseg_fma_half_clamp01(float a, float b) {
    return __builtin_amdgcn_fmed3f(fmaf(a, iseg_opaque_scalar(b), 0.5f), 0.f, 1.f);
}
#ifndef ISEG_GELU_NOPOLY
__device__ __forceinline__ float gelu_poly(float x) {
    const float w = iseg_fma_half_clamp01(x, 1.f / 7.5f) - 0.5f;
    const float t = w * w;
    float q = fmaf(6617.0283203125f, t, -7851.36572265625f);
    q = fmaf(q, t, 3995.16552734375f);
    q = fmaf(q, t, -1156.386962890625f);
    q = fmaf(q, t, 214.41412353515625f);
    q = fmaf(q, t, -27.49638557434082f);
    q = fmaf(q, t, 2.987506628036499f);
    return x * fmaf(w, q, 0.5f);
}
__device__ __forceinline__ float gelu_poly_grad(float x) {
    const float w = iseg_fma_half_clamp01(x, 0.125f) - 0.5f;
    const float t = w * w;
    float r = fmaf(-375307.59375f, t, 481274.125f);
    r = fmaf(r, t, -262435.875f);
    r = fmaf(r, t, 79767.0f);
    r = fmaf(r, t, -14897.58984375f);
    r = fmaf(r, t, 1771.6455078125f);
    r = fmaf(r, t, -132.86141967773438f);
    r = fmaf(r, t, 6.365922927856445f);
    return fmaf(w, r, 0.5f);
}
#else      // A/B build define: the sigmoid forms everywhere (round 4's arithmetic)
__device__ __forceinline__ float gelu_poly(float x) { return gelu_sig(x); }
__device__ __forceinline__ float gelu_poly_grad(float x) {
    float y, d;
    gelu_sig_both(x, y, d);
    return d;
}
#endif

// exact-erf GELU, as keras.activations.gelu(approximate=False)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}
