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
__global__ __launch_bounds__(128) void flash_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                             float* __restrict__ lse2, int64_t items, int T, int heads, int qtiles,
                                                             float scale) {
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
        // training: log2 of the softmax denominator in the exp2 domain of this kernel, rows padded to qtiles*64 (zeros past T)
        if (lse2 && g4 == 0) lse2[bh * ((int64_t)qtiles * FQ) + i] = i < T ? __builtin_fmaf(m[ti], c, __log2f(lt)) : 0.f;
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

// ---------------------------------------------------------------------------------------------------------------------------
// Training.  The forward kernel above also returns L[i] = log2 sum_j exp2(c * s_ij) (c = scale * log2 e); the backward pass
// recomputes the probabilities tile by tile from q, k and L (p = exp2(c * s - L): no running maximum), so nothing of size T x T is
// ever stored.  With D[i] = sum_d dO[i][d] * O[i][d]:   dS = P o (dP - D) * scale,  dP = dO V^T,
//     dV = P^T dO        dK = dS^T Q        dQ = dS K.
// Two kernels, no atomics, fixed summation order (deterministic):
//   * flash_attn_dq_kernel   -- a wavefront owns 64 queries and walks the key tiles in the transposed orientation of the forward
//     kernel (S^T = K Q^T, dP^T = V dO^T; dS^T is the B operand of dQ^T = K^T dS^T, K^T gathered from the K image in LDS);
//   * flash_attn_dkdv_kernel -- a wavefront owns 64 keys and walks the query tiles in the natural orientation (S = Q K^T,
//     dP = dO V^T; P and dS are the B operands of dV^T = dO^T P and dK^T = Q^T dS, dO^T / Q^T gathered from their LDS images).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 frow(const bf16_t* lds, int r0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(lds + (r0 + (lane & 15)) * FSTRIDE + k0 + 8 * (lane >> 4));
}

// D[(b*heads + h) * Tp + i] = sum_d dO[b][i][h*64 + d] * O[b][i][h*64 + d]
__global__ __launch_bounds__(256) void flash_attn_rowdot_kernel(const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
                                                                float* __restrict__ dsum, int64_t B, int T, int heads, int Tp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (b, i, h), h fastest: a wavefront reads contiguous rows
    if (idx >= B * T * heads) return;
    const int h = (int)(idx % heads);
    const int64_t bi = idx / heads;
    const int i = (int)(bi % T);
    const int64_t b = bi / T;
    const bf16_t* o = out + bi * heads * FD + h * FD;
    const bf16_t* d = dout + bi * heads * FD + h * FD;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < FD; k += 8) {
        float a[8], g[8];
        load8<bf16_t>(o + k, a);
        load8<bf16_t>(d + k, g);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_fmaf(a[u], g[u], acc);
    }
    dsum[(b * heads + h) * Tp + i] = acc;
}

__global__ __launch_bounds__(128) void flash_attn_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                            const float* __restrict__ lse2, const float* __restrict__ dsum,
                                                            bf16_t* __restrict__ dqkv, int64_t items, int T, int heads, int qtiles,
                                                            float scale) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bf16_t* Kl = reinterpret_cast<bf16_t*>(fsm) + (size_t)wv * (FK * FSTRIDE);
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
    const bf16_t* dob = dout + b * T * C + h * FD;
    const int q0 = qt * FQ;
    const int jl = lane & 15, g4 = (lane >> 4) * 4;
    const float c = scale * 1.4426950408889634f;
    const int vrow = lane >> 3, vcol = (lane & 7) * 8;

    bf16x8 aq[4][2], ado[4][2], vf[4][2], kr[8];
    float L[4], Dd[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            aq[ti][ks] = fg(qb, ld, q0 + ti * 16 + jl, T, ks * 32, lane);
            ado[ti][ks] = fg(dob, C, q0 + ti * 16 + jl, T, ks * 32, lane);
        }
        const int64_t li = bh * ((int64_t)qtiles * FQ) + q0 + ti * 16 + jl;
        L[ti] = lse2[li];
        Dd[ti] = dsum[li];
    }
    auto fetch = [&](int k0) {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) vf[tj][ks] = fg(vb, ld, k0 + tj * 16 + jl, T, ks * 32, lane);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int row = k0 + vrow + 8 * cc;
            kr[cc] = (row < T) ? *reinterpret_cast<const bf16x8*>(kb + (int64_t)row * ld + vcol) : fzero8();
        }
    };
    fetch(0);
    f32x4 dq[4][4];      // [td][ti]: dQ[i = ti*16 + jl][d = td*16 + g4 + r]
#pragma unroll
    for (int td = 0; td < 4; ++td)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) dq[td][ti] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < T; k0 += FK) {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) *reinterpret_cast<bf16x8*>(Kl + (vrow + 8 * cc) * FSTRIDE + vcol) = kr[cc];
        __builtin_amdgcn_wave_barrier();
        f32x4 s[4][4], dp[4][4];      // [tj][ti]: key k0 + tj*16 + g4 + r, query q0 + ti*16 + jl
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) s[tj][ti] = dp[tj][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const bf16x8 kf = frow(Kl, tj * 16, ks * 32, lane);
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    s[tj][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, aq[ti][ks], s[tj][ti], 0, 0, 0);
                    dp[tj][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[tj][ks], ado[ti][ks], dp[tj][ti], 0, 0, 0);
                }
            }
        const bool ragged = k0 + FK > T;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[tj][ti][r], c, -L[ti]));
                    if (ragged && k0 + tj * 16 + g4 + r >= T) p = 0.f;
                    s[tj][ti][r] = p * (dp[tj][ti][r] - Dd[ti]) * scale;
                }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bds[4], akt[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bds[ti][r] = (bf16_t)s[2 * ks][ti][r];
                    bds[ti][4 + r] = (bf16_t)s[2 * ks + 1][ti][r];
                }
#pragma unroll
            for (int td = 0; td < 4; ++td) akt[td] = fvt(Kl, 32 * ks, 32 * ks + 16, td * 16, lane);
#pragma unroll
            for (int td = 0; td < 4; ++td)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) dq[td][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(akt[td], bds[ti], dq[td][ti], 0, 0, 0);
        }
        if (k0 + FK < T) fetch(k0 + FK);
        __builtin_amdgcn_wave_barrier();
    }
    bf16_t* dqb = dqkv + b * T * ld + h * FD;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const int i = q0 + ti * 16 + jl;
        if (i < T) {
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                bf16x4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (bf16_t)dq[td][ti][r];
                *reinterpret_cast<bf16x4*>(dqb + (int64_t)i * ld + td * 16 + g4) = w;
            }
        }
    }
}

__global__ __launch_bounds__(128) void flash_attn_dkdv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                              const float* __restrict__ lse2, const float* __restrict__ dsum,
                                                              bf16_t* __restrict__ dqkv, int64_t items, int T, int heads, int qtiles,
                                                              float scale) {
    constexpr int IQ = 32;      // queries per inner step (one MFMA k-step of the dV / dK products): keeps the kernel inside 512 VGPRs
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bf16_t* Ql = reinterpret_cast<bf16_t*>(fsm) + (size_t)wv * (2 * IQ * FSTRIDE);
    bf16_t* Ol = Ql + IQ * FSTRIDE;
    const int64_t item = (int64_t)blockIdx.x * 2 + wv;
    if (item >= items) return;
    const int kt = (int)(item % qtiles);
    const int64_t bh = item / qtiles;
    const int h = (int)(bh % heads);
    const int64_t b = bh / heads;
    const int C = heads * FD;
    const int64_t ld = 3 * C;
    const bf16_t* qb = qkv + b * T * ld + h * FD;
    const bf16_t* kb = qb + C;
    const bf16_t* vb = qb + 2 * C;
    const bf16_t* dob = dout + b * T * C + h * FD;
    const float* Lb = lse2 + bh * ((int64_t)qtiles * FQ);
    const float* Db = dsum + bh * ((int64_t)qtiles * FQ);
    const int j0 = kt * FK;
    const int jl = lane & 15, g4 = (lane >> 4) * 4;
    const float c = scale * 1.4426950408889634f;
    const int vrow = lane >> 3, vcol = (lane & 7) * 8;

    bf16x8 kfB[4][2], vfB[4][2], qr[4], dor[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kfB[tj][ks] = fg(kb, ld, j0 + tj * 16 + jl, T, ks * 32, lane);
            vfB[tj][ks] = fg(vb, ld, j0 + tj * 16 + jl, T, ks * 32, lane);
        }
    auto fetch = [&](int i0) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int row = i0 + vrow + 8 * cc;
            const bool ok = row < T;
            qr[cc] = ok ? *reinterpret_cast<const bf16x8*>(qb + (int64_t)row * ld + vcol) : fzero8();
            dor[cc] = ok ? *reinterpret_cast<const bf16x8*>(dob + (int64_t)row * C + vcol) : fzero8();
        }
    };
    fetch(0);
    f32x4 dk[4][4], dv[4][4];      // [td][tj]: key j0 + tj*16 + jl, head column td*16 + g4 + r
#pragma unroll
    for (int td = 0; td < 4; ++td)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) dk[td][tj] = dv[td][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int i0 = 0; i0 < T; i0 += IQ) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            *reinterpret_cast<bf16x8*>(Ql + (vrow + 8 * cc) * FSTRIDE + vcol) = qr[cc];
            *reinterpret_cast<bf16x8*>(Ol + (vrow + 8 * cc) * FSTRIDE + vcol) = dor[cc];
        }
        __builtin_amdgcn_wave_barrier();
        if (i0 + IQ < T) fetch(i0 + IQ);      // next step's rows fly during this step's MFMAs
        f32x4 s[2][4], dp[2][4];      // [ti][tj]: query i0 + ti*16 + g4 + r, key j0 + tj*16 + jl
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) s[ti][tj] = dp[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const bf16x8 qf = frow(Ql, ti * 16, ks * 32, lane), df = frow(Ol, ti * 16, ks * 32, lane);
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    s[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kfB[tj][ks], s[ti][tj], 0, 0, 0);
                    dp[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vfB[tj][ks], dp[ti][tj], 0, 0, 0);
                }
            }
        // rows past T hold zero q / dO (their terms vanish), columns past T are never stored: no masking needed
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const float4 Lr = *reinterpret_cast<const float4*>(Lb + i0 + ti * 16 + g4);
            const float4 Dr = *reinterpret_cast<const float4*>(Db + i0 + ti * 16 + g4);
            const float Lv[4] = {Lr.x, Lr.y, Lr.z, Lr.w}, Dv[4] = {Dr.x, Dr.y, Dr.z, Dr.w};
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ti][tj][r], c, -Lv[r]));
                    s[ti][tj][r] = p;
                    dp[ti][tj][r] = p * (dp[ti][tj][r] - Dv[r]) * scale;
                }
        }
        {
            bf16x8 bp[4], bds[4], adot[4], aqt[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bp[tj][r] = (bf16_t)s[0][tj][r];
                    bp[tj][4 + r] = (bf16_t)s[1][tj][r];
                    bds[tj][r] = (bf16_t)dp[0][tj][r];
                    bds[tj][4 + r] = (bf16_t)dp[1][tj][r];
                }
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                adot[td] = fvt(Ol, 0, 16, td * 16, lane);
                aqt[td] = fvt(Ql, 0, 16, td * 16, lane);
            }
#pragma unroll
            for (int td = 0; td < 4; ++td)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    dv[td][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adot[td], bp[tj], dv[td][tj], 0, 0, 0);
                    dk[td][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aqt[td], bds[tj], dk[td][tj], 0, 0, 0);
                }
        }
        __builtin_amdgcn_wave_barrier();
    }
    bf16_t* dkb = dqkv + b * T * ld + C + h * FD;
    bf16_t* dvb = dkb + C;
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        const int j = j0 + tj * 16 + jl;
        if (j < T) {
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                bf16x4 wk, wvv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    wk[r] = (bf16_t)dk[td][tj][r];
                    wvv[r] = (bf16_t)dv[td][tj][r];
                }
                *reinterpret_cast<bf16x4*>(dkb + (int64_t)j * ld + td * 16 + g4) = wk;
                *reinterpret_cast<bf16x4*>(dvb + (int64_t)j * ld + td * 16 + g4) = wvv;
            }
        }
    }
}

}  // namespace

extern "C" int iseg_attention_fwd_supported(int head_dim, int dtype) { return (head_dim == FD && dtype == ISEG_BF16) ? 1 : 0; }

static int launch_attention_fwd(const void* qkv, void* out, float* lse2, int64_t batch, int T, int heads, int head_dim, float scale,
                                int dtype, hipStream_t stream, const char* who) {
    ISEG_REQUIRE(qkv && out && batch > 0 && T > 0 && heads > 0, "%s: bad arguments", who);
    ISEG_REQUIRE(iseg_attention_fwd_supported(head_dim, dtype), "%s: needs bf16 and head_dim 64 (got %d)", who, head_dim);
    ISEG_REQUIRE(((uintptr_t)qkv | (uintptr_t)out) % 16 == 0, "%s: operands must be 16-byte aligned", who);
    const int qtiles = (T + FQ - 1) / FQ;
    const int64_t items = batch * heads * qtiles;
    const size_t lds = (size_t)2 * FK * FSTRIDE * sizeof(bf16_t);
    hipLaunchKernelGGL(flash_attn_fwd_kernel, dim3((unsigned)ceil_div64(items, 2)), dim3(128), lds, stream, (const bf16_t*)qkv,
                       (bf16_t*)out, lse2, items, T, heads, qtiles, scale);
    return iseg_check_launch(who);
}

extern "C" int iseg_attention_fwd(const void* qkv, void* out, int64_t batch, int T, int heads, int head_dim, float scale, int dtype,
                                  hipStream_t stream) {
    return launch_attention_fwd(qkv, out, nullptr, batch, T, heads, head_dim, scale, dtype, stream, "iseg_attention_fwd");
}

extern "C" size_t iseg_attention_lse_elems(int64_t batch, int T, int heads) {
    return (size_t)batch * heads * (size_t)((T + FQ - 1) / FQ * FQ);
}

extern "C" int iseg_attention_fwd_train(const void* qkv, void* out, float* lse2, int64_t batch, int T, int heads, int head_dim,
                                        float scale, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(lse2 && (uintptr_t)lse2 % 16 == 0, "iseg_attention_fwd_train: lse2 must be a 16-byte aligned buffer");
    return launch_attention_fwd(qkv, out, lse2, batch, T, heads, head_dim, scale, dtype, stream, "iseg_attention_fwd_train");
}

extern "C" size_t iseg_attention_bwd_workspace_bytes(int64_t batch, int T, int heads) {
    return iseg_attention_lse_elems(batch, T, heads) * sizeof(float);
}

extern "C" int iseg_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse2, void* dqkv, int64_t batch, int T,
                                  int heads, int head_dim, float scale, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(qkv && out && dout && lse2 && dqkv && batch > 0 && T > 0 && heads > 0, "iseg_attention_bwd: bad arguments");
    ISEG_REQUIRE(iseg_attention_fwd_supported(head_dim, dtype), "iseg_attention_bwd: needs bf16 and head_dim 64 (got %d)", head_dim);
    ISEG_REQUIRE(((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv | (uintptr_t)lse2) % 16 == 0,
                 "iseg_attention_bwd: operands must be 16-byte aligned");
    const size_t need = iseg_attention_bwd_workspace_bytes(batch, T, heads);
    if (!ws || ws_bytes < need || (uintptr_t)ws % 16) {
        iseg_set_error("iseg_attention_bwd: needs %zu aligned workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    float* dsum = (float*)ws;
    const int qtiles = (T + FQ - 1) / FQ;
    const int Tp = qtiles * FQ;
    const int64_t items = batch * heads * qtiles;
    hipMemsetAsync(dsum, 0, need, stream);      // rows T .. Tp-1 are read (as zeros) by the tile loops
    hipLaunchKernelGGL(flash_attn_rowdot_kernel, dim3((unsigned)ceil_div64(batch * T * heads, 256)), dim3(256), 0, stream,
                       (const bf16_t*)out, (const bf16_t*)dout, dsum, batch, T, heads, Tp);
    hipLaunchKernelGGL(flash_attn_dq_kernel, dim3((unsigned)ceil_div64(items, 2)), dim3(128), (size_t)2 * FK * FSTRIDE * sizeof(bf16_t),
                       stream, (const bf16_t*)qkv, (const bf16_t*)dout, lse2, dsum, (bf16_t*)dqkv, items, T, heads, qtiles, scale);
    hipLaunchKernelGGL(flash_attn_dkdv_kernel, dim3((unsigned)ceil_div64(items, 2)), dim3(128),
                       (size_t)2 * 2 * 32 * FSTRIDE * sizeof(bf16_t), stream, (const bf16_t*)qkv, (const bf16_t*)dout, lse2, dsum,
                       (bf16_t*)dqkv, items, T, heads, qtiles, scale);
    return iseg_check_launch("iseg_attention_bwd");
}
