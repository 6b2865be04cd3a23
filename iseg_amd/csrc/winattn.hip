// Fused window attention for Swin (backbones/swin.py:117-167, WindowAttention.call) on gfx950:
//   O = softmax(scale * Q K^T + bias[h] + mask[w]) V      per (window b_, head h), T = ws*ws <= 64 tokens, head_dim = 32, bf16.
// One wavefront owns one (window, head): S = Q K^T is 4x4 MFMA tiles of v_mfma_f32_16x16x32_bf16 (K = head_dim = one step), the
// softmax runs on the accumulators (row reductions = in-lane over the 4 column tiles + 4 butterfly steps inside each 16-lane
// group), P goes through a wave-private LDS image to become the A operand of P V.  Nothing of size T x T touches HBM (the
// strided-batch GEMM route wrote and re-read S / P and padded every 49-row problem to a 128-row tile).
// Backward recomputes S and P (flash style), forms dP = dO V^T, dS = P (dP - rowsum(dP P)), dV = P^T dO, dQ = scale dS K,
// dK = scale dS^T Q; transposed operands come from LDS through ds_read_b64_tr_b16.  The relative-position-bias gradient
// sum_windows dS is accumulated in registers by persistent wavefronts (one head each) and reduced afterwards in a fixed order.
//
// MFMA fragment conventions (16x16x32 bf16): an A (or B) fragment = lane holds 8 consecutive k = 8*(lane>>4) .. +7 of row (or
// column) lane&15; the accumulator holds D[(lane>>4)*4 + r][lane&15], r = 0..3.
#include "common.h"
#include "iseg_hip.h"

#include <float.h>

namespace {

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

constexpr int WD = 32;            // head dim
constexpr int WT = 64;            // padded tokens
constexpr int PS_STRIDE = 72;     // P / dS image [64][72] bf16 (144-B rows: 16-B aligned, 4-bank skew)
constexpr int QK_STRIDE = 40;     // Q / K / V / dO images [64][40] bf16 (80-B rows)

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f;
    return v;
}

// fragment straight from global memory: row `row` (token) of a [T][ld] matrix, columns col0 + 8*(lane>>4) .. +7
__device__ __forceinline__ bf16x8 gfrag(const bf16_t* __restrict__ base, int64_t ld, int row, int T, int col0, int lane) {
    if (row >= T) return zero8();
    return *reinterpret_cast<const bf16x8*>(base + (int64_t)row * ld + col0 + 8 * (lane >> 4));
}

// fragment from a row-major LDS image [k][stride] read TRANSPOSED: lane receives k0 + 8*(lane>>4) .. +7 of column r0 + (lane&15)
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* lds, int stride, int k0, int r0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const bf16_t* a0 = lds + (k0 + 8 * g + q) * stride + r0 + 4 * p;
    const bf16_t* a1 = a0 + 4 * stride;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a1));
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = lo[i];
        f[4 + i] = hi[i];
    }
    return f;
}

// fragment from a row-major LDS image read along the rows: row r0 + (lane&15), k0 + 8*(lane>>4) .. +7   (one ds_read_b128)
__device__ __forceinline__ bf16x8 row_frag(const bf16_t* lds, int stride, int r0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(lds + (r0 + (lane & 15)) * stride + k0 + 8 * (lane >> 4));
}

// max / sum over the 16 lanes that share (lane >> 4), on the DPP cross-lane network (quad_perm xor 1, xor 2, then half-row and
// row mirrors -- after the two quad steps every lane of a quad holds the quad's value, so the mirrors act as xor 4 / xor 8)
__device__ __forceinline__ float dpp_f(float v, int ctrl) {
    switch (ctrl) {   // dpp_ctrl must be a literal
        case 0: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
        default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
    }
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f(v, 0));
    v = fmaxf(v, dpp_f(v, 1));
    v = fmaxf(v, dpp_f(v, 2));
    return fmaxf(v, dpp_f(v, 3));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f(v, 0);
    v += dpp_f(v, 1);
    v += dpp_f(v, 2);
    return v + dpp_f(v, 3);
}

// scores -> probabilities on the accumulator tiles; s[ti][tj][r] holds S[ti*16 + (lane>>4)*4 + r][tj*16 + (lane&15)].
// `tab` is the [64][64] fp32 additive table of this (mask window, head): bias + mask inside T x T, -FLT_MAX outside, so the
// 64 table loads of a lane carry no bounds logic (the un-padded version spent its time in 128 guarded loads and 150 branches).
__device__ __forceinline__ void softmax_tiles(f32x4 (&s)[4][4], int lane, float scale, const float* __restrict__ tab) {
    const int jl = lane & 15, ib = (lane >> 4) * 4;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* trow = tab + (ti * 16 + ib + r) * WT + jl;
            float v[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) v[tj] = fmaf(s[ti][tj][r], scale, trow[tj * 16]);
            const float mx = row16_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) v[tj] = __expf(v[tj] - mx);
            const float inv = __frcp_rn(row16_sum((v[0] + v[1]) + (v[2] + v[3])));
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) s[ti][tj][r] = v[tj] * inv;
        }
    }
}

// accumulator tiles (fp32) -> bf16 LDS image [64][PS_STRIDE]
__device__ __forceinline__ void tiles_to_lds(const f32x4 (&s)[4][4], bf16_t* img, int lane, float mul) {
    const int jl = lane & 15, ib = (lane >> 4) * 4;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(ti * 16 + ib + r) * PS_STRIDE + tj * 16 + jl] = (bf16_t)(s[ti][tj][r] * mul);
}

// [64 x 32] accumulator (4 x 2 tiles) -> global rows of a [T][ld] matrix at column col0, staged through a [64][QK_STRIDE] LDS image
__device__ __forceinline__ void store_64x32(const f32x4 (&a)[4][2], bf16_t* img, bf16_t* __restrict__ dst, int64_t ld, int col0, int T,
                                            int lane) {
    const int dl = lane & 15, ib = (lane >> 4) * 4;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int td = 0; td < 2; ++td)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(ti * 16 + ib + r) * QK_STRIDE + td * 16 + dl] = (bf16_t)a[ti][td][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // 64 rows x 4 chunks of 8 = 256 chunks, 4 per lane
        const int chunk = lane + c * 64;
        const int row = chunk >> 2, k = (chunk & 3) * 8;
        if (row < T) *reinterpret_cast<bf16x8*>(dst + (int64_t)row * ld + col0 + k) = *reinterpret_cast<const bf16x8*>(img + row * QK_STRIDE + k);
    }
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------------------------
// forward: 4 wavefronts per workgroup, one (window, head) each
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void win_attn_fwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ tab,
                                                           bf16_t* __restrict__ out, int64_t pairs, int T, int heads, int nW,
                                                           float scale) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bf16_t* Pl = reinterpret_cast<bf16_t*>(wsm) + (size_t)wv * (WT * PS_STRIDE + WT * QK_STRIDE);
    bf16_t* Vl = Pl + WT * PS_STRIDE;
    const int64_t pair = (int64_t)blockIdx.x * 4 + wv;
    if (pair >= pairs) return;
    const int C = heads * WD;
    const int64_t ld = 3 * C;
    const int64_t b = pair / heads;
    const int h = (int)(pair % heads);
    const bf16_t* qb = qkv + b * T * ld + h * WD;
    const bf16_t* kb = qb + C;
    const bf16_t* vb = qb + 2 * C;

    // V -> LDS (row-major) while the score MFMAs run
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int chunk = lane + c * 64;
        const int row = chunk >> 2, k = (chunk & 3) * 8;
        bf16x8 v = zero8();
        if (row < T) v = *reinterpret_cast<const bf16x8*>(vb + (int64_t)row * ld + k);
        *reinterpret_cast<bf16x8*>(Vl + row * QK_STRIDE + k) = v;
    }
    bf16x8 aq[4], bk[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        aq[t] = gfrag(qb, ld, t * 16 + (lane & 15), T, 0, lane);
        bk[t] = gfrag(kb, ld, t * 16 + (lane & 15), T, 0, lane);
    }
    f32x4 s[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) s[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ti], bk[tj], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    softmax_tiles(s, lane, scale, tab + ((int64_t)(b % nW) * heads + h) * (WT * WT));
    tiles_to_lds(s, Pl, lane, 1.0f);
    __builtin_amdgcn_wave_barrier();
    f32x4 o[4][2];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int td = 0; td < 2; ++td) o[ti][td] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ap[4], bv[2];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) ap[ti] = row_frag(Pl, PS_STRIDE, ti * 16, ks * 32, lane);
#pragma unroll
        for (int td = 0; td < 2; ++td) bv[td] = tr_frag(Vl, QK_STRIDE, ks * 32, td * 16, lane);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int td = 0; td < 2; ++td) o[ti][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[ti], bv[td], o[ti][td], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    store_64x32(o, Vl, out + b * T * C, C, h * WD, T, lane);
}

// ------------------------------------------------------------------------------------------------------------------------
// backward: 2 wavefronts per workgroup; wave w walks pairs w, w + W, ... (W = total waves, a multiple of `heads`, so the head is
// fixed per wave and the bias gradient can stay in registers until the end)
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void win_attn_bwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ tab,
                                                           const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                           float* __restrict__ dbias_part, int64_t pairs, int T, int heads, int nW,
                                                           float scale) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int PER_WAVE = WT * PS_STRIDE + 3 * WT * QK_STRIDE;
    bf16_t* PSl = reinterpret_cast<bf16_t*>(wsm) + (size_t)wv * PER_WAVE;   // P, then dS
    bf16_t* Ql = PSl + WT * PS_STRIDE;
    bf16_t* Kl = Ql + WT * QK_STRIDE;
    bf16_t* Dl = Kl + WT * QK_STRIDE;                                        // dO
    const int64_t W = (int64_t)gridDim.x * 2;
    const int64_t w = (int64_t)blockIdx.x * 2 + wv;
    const int C = heads * WD;
    const int64_t ld = 3 * C;
    const int h = (int)(w % heads);
    const int jl = lane & 15, ib = (lane >> 4) * 4;

    f32x4 db[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) db[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int64_t pair = w; pair < pairs; pair += W) {
        const int64_t b = pair / heads;   // pair % heads == h because W % heads == 0
        const bf16_t* qb = qkv + b * T * ld + h * WD;
        const bf16_t* kb = qb + C;
        const bf16_t* vb = qb + 2 * C;
        const bf16_t* dob = dout + b * T * C + h * WD;
        bf16x8 aq[4], bk[4], ad[4], bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = t * 16 + (lane & 15);
            aq[t] = gfrag(qb, ld, row, T, 0, lane);
            bk[t] = gfrag(kb, ld, row, T, 0, lane);
            bv[t] = gfrag(vb, ld, row, T, 0, lane);
            ad[t] = gfrag(dob, C, row, T, 0, lane);
            // the same 16-byte pieces, row-major, for the transposed operand reads further down
            *reinterpret_cast<bf16x8*>(Ql + row * QK_STRIDE + 8 * (lane >> 4)) = aq[t];
            *reinterpret_cast<bf16x8*>(Kl + row * QK_STRIDE + 8 * (lane >> 4)) = bk[t];
            *reinterpret_cast<bf16x8*>(Dl + row * QK_STRIDE + 8 * (lane >> 4)) = ad[t];
        }
        f32x4 s[4][4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                s[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ti], bk[tj], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        softmax_tiles(s, lane, scale, tab + ((int64_t)(b % nW) * heads + h) * (WT * WT));
        tiles_to_lds(s, PSl, lane, 1.0f);     // P (bf16) for dV = P^T dO
        // dS = P * (dP - sum_j dP P), one 16-row band of dP = dO V^T at a time (keeps the live accumulators at 64 + 16 + 64)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            f32x4 dp[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                dp[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad[ti], bv[tj], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float dot = 0.f;
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) dot = fmaf(dp[tj][r], s[ti][tj][r], dot);
                dot = row16_sum(dot);
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    const float ds = s[ti][tj][r] * (dp[tj][r] - dot);
                    s[ti][tj][r] = ds;
                    db[ti][tj][r] += ds;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // dV[j][dd] = sum_i P[i][j] dO[i][dd]
        f32x4 acc[4][2];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int td = 0; td < 2; ++td) acc[tj][td] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ap[4], bd[2];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) ap[tj] = tr_frag(PSl, PS_STRIDE, ks * 32, tj * 16, lane);
#pragma unroll
            for (int td = 0; td < 2; ++td) bd[td] = tr_frag(Dl, QK_STRIDE, ks * 32, td * 16, lane);
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int td = 0; td < 2; ++td) acc[tj][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[tj], bd[td], acc[tj][td], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        tiles_to_lds(s, PSl, lane, scale);    // scale * dS (bf16) replaces P
        store_64x32(acc, Dl, dqkv + b * T * ld + 2 * C, ld, h * WD, T, lane);     // dV (the dO image is free now)
        // dQ[i][dd] = sum_j dS[i][j] K[j][dd];  dK[j][dd] = sum_i dS[i][j] Q[i][dd]
        // (dK first: it reads the Q image, which then becomes the staging buffer of dK's own store; dQ reads the K image)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int td = 0; td < 2; ++td) acc[t][td] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 at[4], bqq[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) at[t] = tr_frag(PSl, PS_STRIDE, ks * 32, t * 16, lane);     // rows j, k = i
#pragma unroll
            for (int td = 0; td < 2; ++td) bqq[td] = tr_frag(Ql, QK_STRIDE, ks * 32, td * 16, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int td = 0; td < 2; ++td) acc[t][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[t], bqq[td], acc[t][td], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        store_64x32(acc, Ql, dqkv + b * T * ld + C, ld, h * WD, T, lane);        // dK
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int td = 0; td < 2; ++td) acc[t][td] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ar[4], bkk[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) ar[t] = row_frag(PSl, PS_STRIDE, t * 16, ks * 32, lane);    // rows i, k = j
#pragma unroll
            for (int td = 0; td < 2; ++td) bkk[td] = tr_frag(Kl, QK_STRIDE, ks * 32, td * 16, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int td = 0; td < 2; ++td) acc[t][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[t], bkk[td], acc[t][td], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        store_64x32(acc, Kl, dqkv + b * T * ld, ld, h * WD, T, lane);            // dQ
    }
    // bias-gradient partial of this wave: [W / heads][heads][T][T]
    float* part = dbias_part + ((w / heads) * heads + h) * (int64_t)T * T;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + ib + r, j = tj * 16 + jl;
                if (i < T && j < T) part[i * T + j] = db[ti][tj][r];
            }
}

// tab[w][h][64][64] = bias[h][i][j] + mask[w][i][j] inside T x T, -FLT_MAX elsewhere (w < nW; nW = 1 and mask = null when unshifted)
__global__ __launch_bounds__(256) void win_bias_table_kernel(const float* __restrict__ bias, const float* __restrict__ mask,
                                                             float* __restrict__ tab, int T, int heads, int nW) {
    const int64_t total = (int64_t)nW * heads * WT * WT;
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % WT), i = (int)((e / WT) % WT);
        const int h = (int)((e / (WT * WT)) % heads);
        const int w = (int)(e / ((int64_t)WT * WT * heads));
        float v = -FLT_MAX;
        if (i < T && j < T) {
            v = bias[((int64_t)h * T + i) * T + j];
            if (mask) v += mask[((int64_t)w * T + i) * T + j];
        }
        tab[e] = v;
    }
}

static int bwd_waves(int64_t pairs, int heads) {
    // about 6 waves per CU resident (LDS bound), rounded to a multiple of 2 * heads so that every wave keeps one head
    int64_t w = 256 * 6;
    if (w > pairs) w = pairs;
    const int64_t unit = 2 * (int64_t)heads;
    w = (w / unit) * unit;
    if (w < unit) w = unit;
    return (int)w;
}

}  // namespace

extern "C" int iseg_window_attention_supported(int T, int head_dim, int dtype) {
    return (T > 0 && T <= 64 && head_dim == 32 && dtype == ISEG_BF16) ? 1 : 0;
}

extern "C" int iseg_window_attention_table(const float* bias, const float* mask, float* table, int T, int heads, int mask_windows,
                                           hipStream_t stream) {
    ISEG_REQUIRE(bias && table && T > 0 && T <= WT && heads > 0, "iseg_window_attention_table: bad arguments");
    const int nW = mask ? mask_windows : 1;
    ISEG_REQUIRE(nW > 0, "iseg_window_attention_table: mask needs its window count");
    const int64_t total = (int64_t)nW * heads * WT * WT;
    int64_t blocks = ceil_div64(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(win_bias_table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, bias, mask, table, T, heads, nW);
    return iseg_check_launch("iseg_window_attention_table");
}

extern "C" int iseg_window_attention_fwd(const void* qkv, const float* table, void* out, int64_t windows, int T, int heads,
                                         int table_windows, float scale, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(qkv && table && out && windows > 0 && heads > 0 && table_windows > 0, "iseg_window_attention_fwd: bad arguments");
    ISEG_REQUIRE(iseg_window_attention_supported(T, 32, dtype), "iseg_window_attention_fwd: needs bf16, T <= 64 (got T=%d)", T);
    ISEG_REQUIRE(((uintptr_t)qkv | (uintptr_t)out) % 16 == 0, "iseg_window_attention_fwd: operands must be 16-byte aligned");
    const int64_t pairs = windows * heads;
    const size_t lds = (size_t)4 * (WT * PS_STRIDE + WT * QK_STRIDE) * sizeof(bf16_t);
    hipLaunchKernelGGL(win_attn_fwd_kernel, dim3((unsigned)ceil_div64(pairs, 4)), dim3(256), lds, stream, (const bf16_t*)qkv, table,
                       (bf16_t*)out, pairs, T, heads, table_windows, scale);
    return iseg_check_launch("iseg_window_attention_fwd");
}

extern "C" size_t iseg_window_attention_bwd_workspace_bytes(int64_t windows, int T, int heads) {
    return (size_t)bwd_waves(windows * heads, heads) * (size_t)T * T * sizeof(float);
}

extern "C" int iseg_window_attention_bwd(const void* qkv, const float* table, const void* dout, void* dqkv, float* dbias, int64_t windows,
                                         int T, int heads, int table_windows, float scale, int dtype, void* ws, size_t ws_bytes,
                                         hipStream_t stream) {
    ISEG_REQUIRE(qkv && table && dout && dqkv && dbias && windows > 0 && heads > 0 && table_windows > 0,
                 "iseg_window_attention_bwd: bad arguments");
    ISEG_REQUIRE(iseg_window_attention_supported(T, 32, dtype), "iseg_window_attention_bwd: needs bf16, T <= 64 (got T=%d)", T);
    ISEG_REQUIRE(((uintptr_t)qkv | (uintptr_t)dout | (uintptr_t)dqkv) % 16 == 0, "iseg_window_attention_bwd: operands must be 16-byte aligned");
    const int64_t pairs = windows * heads;
    const int waves = bwd_waves(pairs, heads);
    const size_t need = (size_t)waves * T * T * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_window_attention_bwd: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const size_t lds = (size_t)2 * (WT * PS_STRIDE + 3 * WT * QK_STRIDE) * sizeof(bf16_t);
    hipLaunchKernelGGL(win_attn_bwd_kernel, dim3(waves / 2), dim3(128), lds, stream, (const bf16_t*)qkv, table, (const bf16_t*)dout,
                       (bf16_t*)dqkv, (float*)ws, pairs, T, heads, table_windows, scale);
    // dbias[h][i][j] = sum over the waves / heads partial groups, in group order
    const int64_t n = (int64_t)heads * T * T;
    launch_reduce_rows((const float*)ws, waves / heads, n, 0, 1, n, dbias, nullptr, n, 0, 1.f, 0, stream);
    return iseg_check_launch("iseg_window_attention_bwd");
}
