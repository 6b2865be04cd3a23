// Forward-only fused global self-attention for inference (keras MultiHeadAttention inside backbones/vit.py:142-147,166 and the
// sliding-window driver of core_inference.py): O = softmax(scale * Q K^T) V for packed qkv [B, T, 3*heads*64] bf16, head_dim 64,
// any T.  One wavefront owns 64 query rows of one (sample, head) and walks the keys in tiles of 64 with the online-softmax
// recurrence (running row max m and row sum l, accumulator rescaled by exp(m_old - m_new)), so the T x T score matrix -- 12 x 1025
// x 1025 per ViT-B window, written and re-read twice by the GEMM + softmax route -- never exists.  MFMA v_mfma_f32_16x16x32_bf16
// for K Q^T (two k-steps over the 64-wide head) and V^T P^T (two k-steps over the 64 keys of a tile); K fragments come straight
// from global memory (16 B per lane, prefetched one tile ahead), V goes through a wave-private LDS image and ds_read_b64_tr_b16,
// P stays in registers.
// Training keeps the materialised route (the probabilities are needed by its backward pass).
#include "common.h"
#include "iseg_hip.h"

#include <float.h>

namespace {

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

constexpr int FD = 64;         // head dim
constexpr int FQ = 64;         // query rows per wavefront
constexpr int FK = 64;         // keys per tile
constexpr int FSTRIDE = 72;    // LDS image row stride (bf16): 144-B rows

__device__ __forceinline__ bf16x8 fzero8() {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f;
    return v;
}
__device__ __forceinline__ bf16x8 fg(const bf16_t* __restrict__ base, int64_t ld, int row, int T, int col0, int lane) {
    if (row >= T) return fzero8();
    return *reinterpret_cast<const bf16x8*>(base + (int64_t)row * ld + col0 + 8 * (lane >> 4));
}
// V^T fragment (A operand of O^T = V^T P^T) for head columns c0 .. c0+15: this lane's eight k slots are the keys
// rowA + 4g + 0..3 and rowB + 4g + 0..3 of the LDS image [key][dd] -- the same key set its S^T accumulators hold.
__device__ __forceinline__ bf16x8 fvt(const bf16_t* lds, int rowA, int rowB, int c0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(lds + (rowA + 4 * g + q) * FSTRIDE + c0 + 4 * p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(lds + (rowB + 4 * g + q) * FSTRIDE + c0 + 4 * p));
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = lo[i];
        f[4 + i] = hi[i];
    }
    return f;
}
__device__ __forceinline__ float fxor(float v, int mask) { return __shfl_xor(v, mask, 64); }

// Transposed formulation: S^T = K Q^T (keys on MFMA rows, queries on columns), so each lane owns ONE query column per 16-wide
// query tile and sixteen of its 64 keys: the row max / row sum are 15 in-lane ops + two cross-group exchanges, the running
// (m, l, alpha) are one scalar per query tile, and the probabilities are already laid out as the B operand of O^T = V^T P^T
// (k slots = the lane's own keys; the V^T fragment is gathered to match) -- P never touches LDS.
__global__ __launch_bounds__(128) void flash_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int64_t items,
                                                             int T, int heads, int qtiles, float scale) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bf16_t* Vl = reinterpret_cast<bf16_t*>(fsm) + (size_t)wv * (FK * FSTRIDE);
    const int64_t item = (int64_t)blockIdx.x * 2 + wv;
    if (item >= items) return;
    const int qt = (int)(item % qtiles);
    const int64_t bh = item / qtiles;
    const int h = (int)(bh % heads);
    const int64_t b = bh / heads;
    const int C = heads * FD;
    const int64_t ld = 3 * C;
    const bf16_t* qb = qkv + b * T * ld + h * FD;
    const bf16_t* kb = qb + C;
    const bf16_t* vb = qb + 2 * C;
    const int q0 = qt * FQ;
    const int jl = lane & 15, g4 = (lane >> 4) * 4;
    const float c = scale * 1.4426950408889634f;
    const int vrow = lane >> 3, vcol = (lane & 7) * 8;      // V staging: chunk = lane + 64*cc -> row vrow + 8*cc

    bf16x8 aq[4][2], kf[4][2], vr[8];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aq[ti][ks] = fg(qb, ld, q0 + ti * 16 + jl, T, ks * 32, lane);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[tj][ks] = fg(kb, ld, tj * 16 + jl, T, ks * 32, lane);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const int row = vrow + 8 * cc;
        vr[cc] = (row < T) ? *reinterpret_cast<const bf16x8*>(vb + (int64_t)row * ld + vcol) : fzero8();
    }

    f32x4 o[4][4];      // [td][ti]: O[i = ti*16 + jl][dd = td*16 + g4 + r]
    float m[4], l[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
        for (int td = 0; td < 4; ++td) o[td][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
        m[ti] = -FLT_MAX;
        l[ti] = 0.f;
    }

    for (int k0 = 0; k0 < T; k0 += FK) {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) *reinterpret_cast<bf16x8*>(Vl + (vrow + 8 * cc) * FSTRIDE + vcol) = vr[cc];
        f32x4 s[4][4];      // [tj][ti]: S[i = ti*16 + jl][key = k0 + tj*16 + g4 + r]
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) s[tj][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) s[tj][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[tj][ks], aq[ti][ks], s[tj][ti], 0, 0, 0);
        // next tile's K fragments and V rows start their trip now; they land while this tile's softmax and P V run
        if (k0 + FK < T) {
            const int n0 = k0 + FK;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) kf[tj][ks] = fg(kb, ld, n0 + tj * 16 + jl, T, ks * 32, lane);
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int row = n0 + vrow + 8 * cc;
                vr[cc] = (row < T) ? *reinterpret_cast<const bf16x8*>(vb + (int64_t)row * ld + vcol) : fzero8();
            }
        }
        if (k0 + FK > T) {      // ragged last tile: keys past T get probability zero
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + tj * 16 + g4 + r >= T) {
#pragma unroll
                        for (int ti = 0; ti < 4; ++ti) s[tj][ti][r] = -FLT_MAX;
                    }
        }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            float mx = s[0][ti][0];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[tj][ti][r]);
            mx = fmaxf(mx, fxor(mx, 16));
            mx = fmaxf(mx, fxor(mx, 32));
            const float mnew = fmaxf(m[ti], mx);
            const float alpha = __builtin_amdgcn_exp2f((m[ti] - mnew) * c);
            const float mc = mnew * c;
            m[ti] = mnew;
            float sum = 0.f;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[tj][ti][r], c, -mc));
                    s[tj][ti][r] = pv;
                    sum += pv;
                }
            l[ti] = l[ti] * alpha + sum;      // per-lane partial over this lane's keys; the four key groups are summed once at the end
#pragma unroll
            for (int td = 0; td < 4; ++td)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[td][ti][r] *= alpha;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bp[4], av[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bp[ti][r] = (bf16_t)s[2 * ks][ti][r];
                    bp[ti][4 + r] = (bf16_t)s[2 * ks + 1][ti][r];
                }
#pragma unroll
            for (int td = 0; td < 4; ++td) av[td] = fvt(Vl, 32 * ks, 32 * ks + 16, td * 16, lane);
#pragma unroll
            for (int td = 0; td < 4; ++td)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) o[td][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[td], bp[ti], o[td][ti], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    bf16_t* ob = out + b * T * C + h * FD;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        float lt = l[ti];
        lt += fxor(lt, 16);
        lt += fxor(lt, 32);
        const float inv = __frcp_rn(lt);
        const int i = q0 + ti * 16 + jl;
        if (i < T) {
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                bf16x4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (bf16_t)(o[td][ti][r] * inv);
                *reinterpret_cast<bf16x4*>(ob + (int64_t)i * C + td * 16 + g4) = w;
            }
        }
    }
}

}  // namespace

extern "C" int iseg_attention_fwd_supported(int head_dim, int dtype) { return (head_dim == FD && dtype == ISEG_BF16) ? 1 : 0; }

extern "C" int iseg_attention_fwd(const void* qkv, void* out, int64_t batch, int T, int heads, int head_dim, float scale, int dtype,
                                  hipStream_t stream) {
    ISEG_REQUIRE(qkv && out && batch > 0 && T > 0 && heads > 0, "iseg_attention_fwd: bad arguments");
    ISEG_REQUIRE(iseg_attention_fwd_supported(head_dim, dtype), "iseg_attention_fwd: needs bf16 and head_dim 64 (got %d)", head_dim);
    ISEG_REQUIRE(((uintptr_t)qkv | (uintptr_t)out) % 16 == 0, "iseg_attention_fwd: operands must be 16-byte aligned");
    const int qtiles = (T + FQ - 1) / FQ;
    const int64_t items = batch * heads * qtiles;
    const size_t lds = (size_t)2 * FK * FSTRIDE * sizeof(bf16_t);
    hipLaunchKernelGGL(flash_attn_fwd_kernel, dim3((unsigned)ceil_div64(items, 2)), dim3(128), lds, stream, (const bf16_t*)qkv,
                       (bf16_t*)out, items, T, heads, qtiles, scale);
    return iseg_check_launch("iseg_attention_fwd");
}
