// Helpers shared by the fused ConvNeXt MLP kernels (mlp_fused.hip, mlp_wgrad.hip).
#pragma once
#include "common.h"

// LayerNormalization(axis = -1) applied to one element with the row's saved statistics -- the expression of layernorm_fwd_kernel
// (norm.hip), so that a kernel which re-forms y2 = LN(y1) while staging its operand sees the values the forward pass used.
__device__ __forceinline__ float iseg_ln_apply(float x, float mean, float rstd, float g, float b) { return (x - mean) * rstd * g + b; }

// Byte offset of the 16-byte fragment (row r < 32, k-half h) inside a 1-KiB piece of the tiled weight images (mlp_fused.hip header: the
// k-half is the outer index, so the offset is 16 x the lane that reads it and a ds_read_b128 of a piece is bank-conflict-free)
__device__ __forceinline__ int mlp_frag_offset(int r, int h) {
#ifdef ISEG_MLP_PIECE_ROWMAJOR
    return r * 32 + h * 16;
#else
    return h * 512 + r * 16;
#endif
}

// LayerNorm in front of the fused MLP kernels (mlp_fused.hip).  gamma == NULL: the kernel's row operand is y2 as before.
struct MlpLayerNorm {
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float* mean = nullptr;      // written by the forward kernel, read by the backward kernels
    float* rstd = nullptr;
    float eps = 0.f;
};

// The fused kernels' row fragments: lane (r = lane & 31, h = lane >> 5) holds channels 16 kk + 8 h .. + 7 of row r for kk < KK, i.e. a row
// is split over the two lanes r and r + 32.
// mlp_layernorm_rows: statistics (biased variance, keras LayerNormalization) + normalisation in place; row < 0 = padding row (no store).
template <int KK> __device__ __forceinline__ void mlp_layernorm_rows(bf16x8 (&yf)[KK], const MlpLayerNorm& ln, int64_t row, int h, int C) {
    float s = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (float)yf[kk][u];
    {
        const unsigned v = __builtin_bit_cast(unsigned, s);
        auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        s = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float d = (float)yf[kk][u] - mean;
            q += d * d;
        }
    {
        const unsigned v = __builtin_bit_cast(unsigned, q);
        auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        q = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
    const float rstd = rsqrtf(q / (float)C + ln.eps);
    if (row >= 0 && h == 0) {
        ln.mean[row] = mean;
        ln.rstd[row] = rstd;
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        float g[8], b[8];
        load8<float>(ln.gamma + 16 * kk + 8 * h, g);
        load8<float>(ln.beta + 16 * kk + 8 * h, b);
#pragma unroll
        for (int u = 0; u < 8; ++u) yf[kk][u] = (bf16_t)iseg_ln_apply((float)yf[kk][u], mean, rstd, g[u], b[u]);
    }
}

// the same normalisation from SAVED statistics (backward kernels)
template <int KK> __device__ __forceinline__ void mlp_layernorm_apply_rows(bf16x8 (&yf)[KK], const MlpLayerNorm& ln, int64_t row, int h) {
    const float mean = ln.mean[row], rstd = ln.rstd[row];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        float g[8], b[8];
        load8<float>(ln.gamma + 16 * kk + 8 * h, g);
        load8<float>(ln.beta + 16 * kk + 8 * h, b);
#pragma unroll
        for (int u = 0; u < 8; ++u) yf[kk][u] = (bf16_t)iseg_ln_apply((float)yf[kk][u], mean, rstd, g[u], b[u]);
    }
}
