// Forward-only fused global self-attention for inference (keras MultiHeadAttention inside backbones/vit.py:142-147,166 and the
// sliding-window driver of core_inference.py): O = softmax(scale * Q K^T) V for packed qkv [B, T, 3*heads*64] bf16, head_dim 64,
// any T.  One wavefront owns 64 query rows of one (sample, head) and walks the keys in tiles of 64 with the online-softmax
// recurrence (running row max m and row sum l, accumulator rescaled by exp(m_old - m_new)), so the T x T score matrix -- 12 x 1025
// x 1025 per ViT-B window, written and re-read twice by the GEMM + softmax route -- never exists.  MFMA v_mfma_f32_16x16x32_bf16
// for Q K^T (two k-steps over the 64-wide head) and P V (two k-steps over the 64 keys of a tile); K fragments come straight from
// global memory (16 B per lane), V goes through a wave-private LDS image and ds_read_b64_tr_b16, P through a second image.
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
__device__ __forceinline__ bf16x8 ftr(const bf16_t* lds, int k0, int r0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const bf16_t* a0 = lds + (k0 + 8 * g + q) * FSTRIDE + r0 + 4 * p;
    const bf16_t* a1 = a0 + 4 * FSTRIDE;
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
__device__ __forceinline__ bf16x8 frow(const bf16_t* lds, int r0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(lds + (r0 + (lane & 15)) * FSTRIDE + k0 + 8 * (lane >> 4));
}
__device__ __forceinline__ float fdpp(float v, int ctrl) {
    switch (ctrl) {
        case 0: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
        case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
        case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
        default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    }
}
__device__ __forceinline__ float f16max(float v) {
    v = fmaxf(v, fdpp(v, 0));
    v = fmaxf(v, fdpp(v, 1));
    v = fmaxf(v, fdpp(v, 2));
    return fmaxf(v, fdpp(v, 3));
}
__device__ __forceinline__ float f16sum(float v) {
    v += fdpp(v, 0);
    v += fdpp(v, 1);
    v += fdpp(v, 2);
    return v + fdpp(v, 3);
}

__global__ __launch_bounds__(128) void flash_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int64_t items,
                                                             int T, int heads, int qtiles, float scale) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bf16_t* Pl = reinterpret_cast<bf16_t*>(fsm) + (size_t)wv * (2 * FQ * FSTRIDE);
    bf16_t* Vl = Pl + FQ * FSTRIDE;
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
    const int jl = lane & 15, ib = (lane >> 4) * 4;

    bf16x8 aq[4][2];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aq[ti][ks] = fg(qb, ld, q0 + ti * 16 + (lane & 15), T, ks * 32, lane);

    f32x4 o[4][4];
    float m[4][4], l[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            o[ti][x] = f32x4{0.f, 0.f, 0.f, 0.f};
            m[ti][x] = -FLT_MAX;
            l[ti][x] = 0.f;
        }

    for (int k0 = 0; k0 < T; k0 += FK) {
        // V tile -> LDS (row-major [key][dd]); 64 rows x 8 chunks of 16 B = 512 chunks, 8 per lane
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int chunk = lane + c * 64;
            const int row = chunk >> 3, k = (chunk & 7) * 8;
            bf16x8 v = fzero8();
            if (k0 + row < T) v = *reinterpret_cast<const bf16x8*>(vb + (int64_t)(k0 + row) * ld + k);
            *reinterpret_cast<bf16x8*>(Vl + row * FSTRIDE + k) = v;
        }
        // S = Q K^T for this key tile
        f32x4 s[4][4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) s[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bk[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) bk[tj] = fg(kb, ld, k0 + tj * 16 + (lane & 15), T, ks * 32, lane);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) s[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ti][ks], bk[tj], s[ti][tj], 0, 0, 0);
        }
        // online softmax on the tile: rows (ti, r) <-> i = ti*16 + ib + r, columns (tj, lane&15) <-> key k0 + tj*16 + jl
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v[4];
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) v[tj] = (k0 + tj * 16 + jl < T) ? s[ti][tj][r] * scale : -FLT_MAX;
                const float mnew = fmaxf(m[ti][r], f16max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]))));
                const float alpha = __expf(m[ti][r] - mnew);
                m[ti][r] = mnew;
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    v[tj] = __expf(v[tj] - mnew);
                    s[ti][tj][r] = v[tj];
                }
                l[ti][r] = l[ti][r] * alpha + f16sum((v[0] + v[1]) + (v[2] + v[3]));
#pragma unroll
                for (int td = 0; td < 4; ++td) o[ti][td][r] *= alpha;
            }
        // P -> LDS image, then O += P V
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) Pl[(ti * 16 + ib + r) * FSTRIDE + tj * 16 + jl] = (bf16_t)s[ti][tj][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ap[4], bv[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) ap[ti] = frow(Pl, ti * 16, ks * 32, lane);
#pragma unroll
            for (int td = 0; td < 4; ++td) bv[td] = ftr(Vl, ks * 32, td * 16, lane);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int td = 0; td < 4; ++td) o[ti][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[ti], bv[td], o[ti][td], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // O / l -> bf16 rows through the P image (64 x 64), then 16-byte coalesced stores
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float inv = __frcp_rn(l[ti][r]);
#pragma unroll
            for (int td = 0; td < 4; ++td) Pl[(ti * 16 + ib + r) * FSTRIDE + td * 16 + jl] = (bf16_t)(o[ti][td][r] * inv);
        }
    __builtin_amdgcn_wave_barrier();
    bf16_t* ob = out + b * T * C + h * FD;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int chunk = lane + c * 64;
        const int row = chunk >> 3, k = (chunk & 7) * 8;
        if (q0 + row < T) *reinterpret_cast<bf16x8*>(ob + (int64_t)(q0 + row) * C + k) = *reinterpret_cast<const bf16x8*>(Pl + row * FSTRIDE + k);
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
    const size_t lds = (size_t)2 * 2 * FQ * FSTRIDE * sizeof(bf16_t);
    hipLaunchKernelGGL(flash_attn_fwd_kernel, dim3((unsigned)ceil_div64(items, 2)), dim3(128), lds, stream, (const bf16_t*)qkv,
                       (bf16_t*)out, items, T, heads, qtiles, scale);
    return iseg_check_launch("iseg_attention_fwd");
}
