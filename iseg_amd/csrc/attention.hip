// Attention support kernels.  The score / context products are strided-batch GEMMs (gemm.hip, one problem per (sample, head));
// this file holds what sits between them and around them:
//   * row softmax over keys with the additive terms of backbones/swin.py:131-158 (relative-position bias [heads,T,T], shift
//     mask [nW,T,T]) and the probability clip of layers/multihead_self_attention.py:138, forward and backward
//   * tf.clip_by_value forward / backward (used when dropout sits between softmax and clip)
//   * row gather with an int32 index (-1 = zero row): tf.pad + tf.roll + window_partition / window_reverse + crop of
//     backbones/swin.py:46-64,258-288 and the 2x2 space-to-depth of PatchMerging (:316-327) are all row permutations
//   * relative-position bias table gather (swin.py:134-142) and its gradient
#include "common.h"
#include "iseg_hip.h"

#include <float.h>

namespace {

// one wave per row; `cols` valid entries, row stride `ld` (>= cols, pad columns are written as zeros)
template <class T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T* __restrict__ s, T* __restrict__ p, int64_t rows, int Tq, int cols,
                                                          int ld, const float* __restrict__ bias, int heads,
                                                          const float* __restrict__ mask, int nW, float clip_lo, float clip_hi) {
    const int lane = threadIdx.x & 63;
    const bool clip = clip_hi > clip_lo;
    for (int64_t r = blockIdx.x * 4ll + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int64_t z = r / Tq;
        const int i = (int)(r % Tq);
        const T* sr = s + r * ld;
        T* pr = p + r * ld;
        const float* br = bias ? bias + ((int64_t)(z % heads) * Tq + i) * cols : nullptr;
        const float* mr = mask ? mask + ((int64_t)((z / heads) % nW) * Tq + i) * cols : nullptr;
        float mx = -FLT_MAX;
        for (int j = lane; j < cols; j += 64) {
            float v = to_f32(sr[j]);
            if (br) v += br[j];
            if (mr) v += mr[j];
            mx = fmaxf(mx, v);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int j = lane; j < cols; j += 64) {
            float v = to_f32(sr[j]);
            if (br) v += br[j];
            if (mr) v += mr[j];
            sum += __expf(v - mx);
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int j = lane; j < ld; j += 64) {
            float out = 0.f;
            if (j < cols) {
                float v = to_f32(sr[j]);
                if (br) v += br[j];
                if (mr) v += mr[j];
                out = __expf(v - mx) * inv;
                if (clip) out = fminf(fmaxf(out, clip_lo), clip_hi);
            }
            pr[j] = from_f32<T>(out);
        }
    }
}

// dS = P * (g - sum_j g_j P_j), g = dP where the clip passed (lo < P < hi as stored), else 0.  In place on dP is allowed.
template <class T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp, T* __restrict__ ds,
                                                          int64_t rows, int cols, int ld, float clip_lo, float clip_hi) {
    const int lane = threadIdx.x & 63;
    const bool clip = clip_hi > clip_lo;
    for (int64_t r = blockIdx.x * 4ll + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const T* pr = p + r * ld;
        const T* dr = dp + r * ld;
        T* or_ = ds + r * ld;
        float dot = 0.f;
        for (int j = lane; j < cols; j += 64) {
            const float pv = to_f32(pr[j]);
            float g = to_f32(dr[j]);
            if (clip && !(pv > clip_lo && pv < clip_hi)) g = 0.f;
            dot = fmaf(g, pv, dot);
        }
        dot = wave_sum(dot);
        for (int j = lane; j < ld; j += 64) {
            float out = 0.f;
            if (j < cols) {
                const float pv = to_f32(pr[j]);
                float g = to_f32(dr[j]);
                if (clip && !(pv > clip_lo && pv < clip_hi)) g = 0.f;
                out = pv * (g - dot);
            }
            or_[j] = from_f32<T>(out);
        }
    }
}

// Register-resident variants: a row is read once and written once.  LPR lanes share a row (64 / LPR rows per wave: rows of 9
// (DCNv3 mask) or 49 (Swin window) no longer leave most of a wave idle), lane l holds columns l, l + LPR, ... (EPL of them).
template <class T, int LPR, int EPL>
__global__ __launch_bounds__(256) void softmax_fwd_reg_kernel(const T* __restrict__ s, T* __restrict__ p, int64_t rows, int Tq, int cols,
                                                              int ld, const float* __restrict__ bias, int heads,
                                                              const float* __restrict__ mask, int nW, float clip_lo, float clip_hi) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, l = lane % LPR, sub = lane / LPR;
    const bool clip = clip_hi > clip_lo;
    for (int64_t r = (blockIdx.x * 4ll + (threadIdx.x >> 6)) * RPW + sub; r < rows; r += (int64_t)gridDim.x * 4 * RPW) {
        const int64_t z = r / Tq;
        const int i = (int)(r % Tq);
        const T* sr = s + r * ld;
        T* pr = p + r * ld;
        const float* br = bias ? bias + ((int64_t)(z % heads) * Tq + i) * cols : nullptr;
        const float* mr = mask ? mask + ((int64_t)((z / heads) % nW) * Tq + i) * cols : nullptr;
        float v[EPL];
        float mx = -FLT_MAX;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int j = l + k * LPR;
            v[k] = -FLT_MAX;
            if (j < cols) {
                float t = to_f32(sr[j]);
                if (br) t += br[j];
                if (mr) t += mr[j];
                v[k] = t;
                mx = fmaxf(mx, t);
            }
        }
#pragma unroll
        for (int o = LPR >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            v[k] = (l + k * LPR < cols) ? __expf(v[k] - mx) : 0.f;
            sum += v[k];
        }
        sum = group_sum(sum, LPR);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int j = l + k * LPR;
            if (j < ld) {
                float out = v[k] * inv;
                if (clip && j < cols) out = fminf(fmaxf(out, clip_lo), clip_hi);
                pr[j] = from_f32<T>(j < cols ? out : 0.f);
            }
        }
    }
}

template <class T, int LPR, int EPL>
__global__ __launch_bounds__(256) void softmax_bwd_reg_kernel(const T* __restrict__ p, const T* __restrict__ dp, T* __restrict__ ds,
                                                              int64_t rows, int cols, int ld, float clip_lo, float clip_hi) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, l = lane % LPR, sub = lane / LPR;
    const bool clip = clip_hi > clip_lo;
    for (int64_t r = (blockIdx.x * 4ll + (threadIdx.x >> 6)) * RPW + sub; r < rows; r += (int64_t)gridDim.x * 4 * RPW) {
        const T* pr = p + r * ld;
        const T* dr = dp + r * ld;
        T* or_ = ds + r * ld;
        float pv[EPL], g[EPL];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int j = l + k * LPR;
            pv[k] = g[k] = 0.f;
            if (j < cols) {
                pv[k] = to_f32(pr[j]);
                g[k] = to_f32(dr[j]);
                if (clip && !(pv[k] > clip_lo && pv[k] < clip_hi)) g[k] = 0.f;
                dot = fmaf(g[k], pv[k], dot);
            }
        }
        dot = group_sum(dot, LPR);
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int j = l + k * LPR;
            if (j < ld) or_[j] = from_f32<T>(j < cols ? pv[k] * (g[k] - dot) : 0.f);
        }
    }
}

// picks (LPR, EPL) for a row length; returns false when the row does not fit the register variants (ld > 64 * 32)
template <class F16, class F64x2, class F64x8, class F64x32>
static bool softmax_dispatch(int ld, F16 f16, F64x2 f64x2, F64x8 f64x8, F64x32 f64x32) {
    if (ld <= 16) f16();
    else if (ld <= 128) f64x2();
    else if (ld <= 512) f64x8();
    else if (ld <= 2048) f64x32();
    else return false;
    return true;
}

template <class T>
__global__ __launch_bounds__(256) void clip_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float lo, float hi) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = from_f32<T>(fminf(fmaxf(to_f32(x[i]), lo), hi));
}
template <class T>
__global__ __launch_bounds__(256) void clip_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int64_t n,
                                                       float lo, float hi) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = to_f32(x[i]);
        dx[i] = (v >= lo && v <= hi) ? dy[i] : from_f32<T>(0.f);
    }
}

// y[r, :] = idx[r] >= 0 ? x[idx[r], :] : 0     (rows of C elements; 16-byte chunks when C*sizeof(T) % 16 == 0)
template <class T, int VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ x, const int32_t* __restrict__ idx, T* __restrict__ y,
                                                          int64_t rows_out, int C) {
    const int chunks = C / VEC;
    const int64_t total = rows_out * chunks;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c = (int)(i % chunks) * VEC;
        const int32_t src = idx[r];
        if (VEC * sizeof(T) == 16) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src >= 0) v = *reinterpret_cast<const float4*>(x + (int64_t)src * C + c);
            *reinterpret_cast<float4*>(y + r * C + c) = v;
        } else {
#pragma unroll
            for (int u = 0; u < VEC; ++u) y[r * C + c + u] = src >= 0 ? x[(int64_t)src * C + c + u] : from_f32<T>(0.f);
        }
    }
}

// y[r, :] = res[r, :] + scale[g] * (idx[r] >= 0 ? x[idx[r], :] : 0),  g = (by_src ? idx[r] : r) / rows_per_group   (res, scale optional; C % 8 == 0)
// -- window reverse + roll back + crop + drop path + skip connection of a Swin block in one pass (backbones/swin.py:264-279: by_src = 0), and
// its gradient towards the window rows (the inverse table, the factor of the SOURCE row's sample: by_src = 1)
template <class T>
__global__ __launch_bounds__(256) void gather_rows_fma_kernel(const T* __restrict__ x, const int32_t* __restrict__ idx, const float* __restrict__ scale,
                                                              int64_t rows_per_group, int by_src, const T* __restrict__ res, T* __restrict__ y,
                                                              int64_t rows_out, int C) {
    const int chunks = C / 8;
    const int64_t total = rows_out * chunks;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c = (int)(i % chunks) * 8;
        const int32_t src = idx[r];
        float v[8], a[8];
        load8<T>(x + (int64_t)(src >= 0 ? src : 0) * C + c, v);      // (unconditional load from a clamped address, selected afterwards)
        if (res) load8<T>(res + r * C + c, a);
        float f = src >= 0 ? 1.f : 0.f;
        if (scale && src >= 0) f = scale[(by_src ? (int64_t)src : r) / rows_per_group];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = res ? fmaf(f, v[u], a[u]) : f * v[u];
        store8<T>(y + r * C + c, v);
    }
}

__global__ __launch_bounds__(256) void relpos_gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ index,
                                                            float* __restrict__ bias, int heads, int TT) {
    const int total = heads * TT;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int h = i / TT, ij = i % TT;
        bias[i] = table[(int64_t)index[ij] * heads + h];
    }
}

// dtable[k, h] (+)= sum over (i,j) with index[i,j] == k of dbias[h, i, j]; one thread per table entry, fixed order
__global__ __launch_bounds__(256) void relpos_scatter_kernel(const float* __restrict__ dbias, int ld, const int32_t* __restrict__ index,
                                                             float* __restrict__ dtable, int entries, int heads, int T,
                                                             int accumulate) {
    const int total = entries * heads;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int k = i / heads, h = i % heads;
        float s = 0.f;
        for (int a = 0; a < T; ++a)
            for (int b = 0; b < T; ++b)
                if (index[a * T + b] == k) s += dbias[((int64_t)h * T + a) * ld + b];
        dtable[i] = accumulate ? dtable[i] + s : s;
    }
}

// Same gradient for the canonical square-window index of backbones/swin.py:93-104 (index[i][j] = (yi-yj+ws-1)*(2ws-1) + (xi-xj+ws-1)):
// entry k = (dy, dx) collects the <= ws*ws pairs with that displacement, enumerated directly instead of scanning all T*T pairs
// (162 us -> a few us per call at T = 49).  The host checks the index table against the formula before choosing this kernel.
__global__ __launch_bounds__(256) void relpos_scatter_ws_kernel(const float* __restrict__ dbias, int ld, float* __restrict__ dtable,
                                                                int ws, int heads, int accumulate) {
    // one wavefront per table entry, lanes over the query positions (one independent load each, summed by a fixed shuffle tree): the one-thread-
    // per-entry form walked its <= 49 pairs as a chain of dependent loads (13.7 us per call, Swin-T: 12 calls per step)
    const int side = 2 * ws - 1, T = ws * ws;
    const int total = side * side * heads;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= total) return;      // wave-uniform
    const int k = i / heads, h = i % heads;
    const int dy = k / side - (ws - 1), dx = k % side - (ws - 1);
    float s = 0.f;
    for (int a = lane; a < T; a += 64) {
        const int yi = a / ws, xi = a % ws;
        const int yj = yi - dy, xj = xi - dx;
        if (yj >= 0 && yj < ws && xj >= 0 && xj < ws) s += dbias[((int64_t)h * T + a) * ld + yj * ws + xj];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) dtable[i] = accumulate ? dtable[i] + s : s;
}

// column sums of a short, very wide matrix (rows = windows, cols = heads*T*ld score entries): lanes along the columns,
// row chunks along grid.y, partials [chunks][cols] reduced afterwards in chunk order
template <class T>
__global__ __launch_bounds__(256) void colsum_wide_partial_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int64_t cols,
                                                                  float* __restrict__ partials) {
    const int64_t c = blockIdx.x * 256ll + threadIdx.x;
    if (c >= cols) return;
    const int64_t rpc = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = blockIdx.y * rpc, r1 = min(rows, r0 + rpc);
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += to_f32(x[r * ldx + c]);
    partials[(int64_t)blockIdx.y * cols + c] = s;
}

static inline int wide_chunks(int64_t rows) {
    int64_t c = ceil_div64(rows, 32);
    if (c > 64) c = 64;
    if (c < 1) c = 1;
    return (int)c;
}

static inline unsigned row_blocks(int64_t rows) {
    int64_t b = ceil_div64(rows, 4);
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (unsigned)b;
}
static inline unsigned ew_blocks(int64_t n) {
    int64_t b = ceil_div64(n, 256);
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int iseg_softmax_rows_fwd(const void* scores, void* probs, int64_t problems, int Tq, int cols, int ld, const float* bias,
                                     int heads, const float* mask, int windows, float clip_lo, float clip_hi, int dtype,
                                     hipStream_t stream) {
    ISEG_REQUIRE(scores && probs && problems > 0 && Tq > 0 && cols > 0 && ld >= cols, "iseg_softmax_rows_fwd: bad arguments");
    ISEG_REQUIRE((!bias && !mask) || heads > 0, "iseg_softmax_rows_fwd: bias / mask need the head count");
    ISEG_REQUIRE(!mask || windows > 0, "iseg_softmax_rows_fwd: mask needs the window count");
    const int64_t rows = problems * Tq;
    const int hd = heads > 0 ? heads : 1, nw = windows > 0 ? windows : 1;
#define SM_FWD(T, LPR, EPL)                                                                                                          \
    hipLaunchKernelGGL((softmax_fwd_reg_kernel<T, LPR, EPL>), dim3(row_blocks(ceil_div64(rows, 64 / LPR))), dim3(256), 0, stream,    \
                       (const T*)scores, (T*)probs, rows, Tq, cols, ld, bias, hd, mask, nw, clip_lo, clip_hi)
    bool done;
    if (dtype == ISEG_BF16)
        done = softmax_dispatch(ld, [&] { SM_FWD(bf16_t, 16, 1); }, [&] { SM_FWD(bf16_t, 64, 2); }, [&] { SM_FWD(bf16_t, 64, 8); },
                                [&] { SM_FWD(bf16_t, 64, 32); });
    else
        done = softmax_dispatch(ld, [&] { SM_FWD(float, 16, 1); }, [&] { SM_FWD(float, 64, 2); }, [&] { SM_FWD(float, 64, 8); },
                                [&] { SM_FWD(float, 64, 32); });
#undef SM_FWD
    if (done) return iseg_check_launch("iseg_softmax_rows_fwd");
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((softmax_fwd_kernel<bf16_t>), dim3(row_blocks(rows)), dim3(256), 0, stream, (const bf16_t*)scores,
                           (bf16_t*)probs, rows, Tq, cols, ld, bias, heads > 0 ? heads : 1, mask, windows > 0 ? windows : 1, clip_lo,
                           clip_hi);
    else
        hipLaunchKernelGGL((softmax_fwd_kernel<float>), dim3(row_blocks(rows)), dim3(256), 0, stream, (const float*)scores,
                           (float*)probs, rows, Tq, cols, ld, bias, heads > 0 ? heads : 1, mask, windows > 0 ? windows : 1, clip_lo,
                           clip_hi);
    return iseg_check_launch("iseg_softmax_rows_fwd");
}

extern "C" int iseg_softmax_rows_bwd(const void* probs, const void* dprobs, void* dscores, int64_t rows, int cols, int ld, float clip_lo,
                                     float clip_hi, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(probs && dprobs && dscores && rows > 0 && cols > 0 && ld >= cols, "iseg_softmax_rows_bwd: bad arguments");
#define SM_BWD(T, LPR, EPL)                                                                                                          \
    hipLaunchKernelGGL((softmax_bwd_reg_kernel<T, LPR, EPL>), dim3(row_blocks(ceil_div64(rows, 64 / LPR))), dim3(256), 0, stream,    \
                       (const T*)probs, (const T*)dprobs, (T*)dscores, rows, cols, ld, clip_lo, clip_hi)
    bool done;
    if (dtype == ISEG_BF16)
        done = softmax_dispatch(ld, [&] { SM_BWD(bf16_t, 16, 1); }, [&] { SM_BWD(bf16_t, 64, 2); }, [&] { SM_BWD(bf16_t, 64, 8); },
                                [&] { SM_BWD(bf16_t, 64, 32); });
    else
        done = softmax_dispatch(ld, [&] { SM_BWD(float, 16, 1); }, [&] { SM_BWD(float, 64, 2); }, [&] { SM_BWD(float, 64, 8); },
                                [&] { SM_BWD(float, 64, 32); });
#undef SM_BWD
    if (done) return iseg_check_launch("iseg_softmax_rows_bwd");
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t>), dim3(row_blocks(rows)), dim3(256), 0, stream, (const bf16_t*)probs,
                           (const bf16_t*)dprobs, (bf16_t*)dscores, rows, cols, ld, clip_lo, clip_hi);
    else
        hipLaunchKernelGGL((softmax_bwd_kernel<float>), dim3(row_blocks(rows)), dim3(256), 0, stream, (const float*)probs,
                           (const float*)dprobs, (float*)dscores, rows, cols, ld, clip_lo, clip_hi);
    return iseg_check_launch("iseg_softmax_rows_bwd");
}

extern "C" int iseg_clip_fwd(const void* x, void* y, int64_t n, float lo, float hi, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && y && n > 0 && hi >= lo, "iseg_clip_fwd: bad arguments");
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((clip_fwd_kernel<bf16_t>), dim3(ew_blocks(n)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n, lo, hi);
    else
        hipLaunchKernelGGL((clip_fwd_kernel<float>), dim3(ew_blocks(n)), dim3(256), 0, stream, (const float*)x, (float*)y, n, lo, hi);
    return iseg_check_launch("iseg_clip_fwd");
}

extern "C" int iseg_clip_bwd(const void* x, const void* dy, void* dx, int64_t n, float lo, float hi, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && dy && dx && n > 0 && hi >= lo, "iseg_clip_bwd: bad arguments");
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((clip_bwd_kernel<bf16_t>), dim3(ew_blocks(n)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy,
                           (bf16_t*)dx, n, lo, hi);
    else
        hipLaunchKernelGGL((clip_bwd_kernel<float>), dim3(ew_blocks(n)), dim3(256), 0, stream, (const float*)x, (const float*)dy,
                           (float*)dx, n, lo, hi);
    return iseg_check_launch("iseg_clip_bwd");
}

extern "C" int iseg_gather_rows(const void* x, const int32_t* idx, void* y, int64_t rows_in, int64_t rows_out, int C, int dtype,
                                hipStream_t stream) {
    ISEG_REQUIRE(x && idx && y && rows_in > 0 && rows_out > 0 && C > 0, "iseg_gather_rows: bad arguments");
    (void)rows_in;   // the caller guarantees idx[r] < rows_in (index tables are built on the host from static geometry)
    const size_t es = dtype_size(dtype);
    const bool vec = ((size_t)C * es) % 16 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0;
    if (dtype == ISEG_BF16) {
        if (vec)
            hipLaunchKernelGGL((gather_rows_kernel<bf16_t, 8>), dim3(ew_blocks(rows_out * (C / 8))), dim3(256), 0, stream,
                               (const bf16_t*)x, idx, (bf16_t*)y, rows_out, C);
        else
            hipLaunchKernelGGL((gather_rows_kernel<bf16_t, 1>), dim3(ew_blocks(rows_out * C)), dim3(256), 0, stream, (const bf16_t*)x,
                               idx, (bf16_t*)y, rows_out, C);
    } else {
        if (vec)
            hipLaunchKernelGGL((gather_rows_kernel<float, 4>), dim3(ew_blocks(rows_out * (C / 4))), dim3(256), 0, stream, (const float*)x,
                               idx, (float*)y, rows_out, C);
        else
            hipLaunchKernelGGL((gather_rows_kernel<float, 1>), dim3(ew_blocks(rows_out * C)), dim3(256), 0, stream, (const float*)x, idx,
                               (float*)y, rows_out, C);
    }
    return iseg_check_launch("iseg_gather_rows");
}

extern "C" int iseg_gather_rows_fma(const void* x, const int32_t* idx, const float* scale, int64_t rows_per_group, int scale_by_source_row,
                                    const void* residual, void* y, int64_t rows_out, int C, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && idx && y && rows_out > 0 && C > 0 && C % 8 == 0, "iseg_gather_rows_fma: bad arguments (C = %d must be a multiple of 8)", C);
    ISEG_REQUIRE(!scale || rows_per_group > 0, "iseg_gather_rows_fma: scale needs rows_per_group > 0");
    ISEG_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0, "iseg_gather_rows_fma: operands must be 16-byte aligned");
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((gather_rows_fma_kernel<bf16_t>), dim3(ew_blocks(rows_out * (C / 8))), dim3(256), 0, stream, (const bf16_t*)x, idx, scale,
                           rows_per_group, scale_by_source_row, (const bf16_t*)residual, (bf16_t*)y, rows_out, C);
    else
        hipLaunchKernelGGL((gather_rows_fma_kernel<float>), dim3(ew_blocks(rows_out * (C / 8))), dim3(256), 0, stream, (const float*)x, idx, scale,
                           rows_per_group, scale_by_source_row, (const float*)residual, (float*)y, rows_out, C);
    return iseg_check_launch("iseg_gather_rows_fma");
}

extern "C" int iseg_relpos_bias_scatter_grad_window(const float* dbias, int ld, float* dtable, int ws, int heads, int accumulate,
                                                    hipStream_t stream) {
    ISEG_REQUIRE(dbias && dtable && ws > 0 && heads > 0 && ld >= ws * ws, "iseg_relpos_bias_scatter_grad_window: bad arguments");
    const int total = (2 * ws - 1) * (2 * ws - 1) * heads;
    hipLaunchKernelGGL(relpos_scatter_ws_kernel, dim3((total + 3) / 4), dim3(256), 0, stream, dbias, ld, dtable, ws, heads, accumulate);
    return iseg_check_launch("iseg_relpos_bias_scatter_grad_window");
}

extern "C" size_t iseg_colsum_wide_workspace_bytes(int64_t rows, int64_t cols) {
    return (size_t)wide_chunks(rows) * (size_t)cols * sizeof(float);
}

extern "C" int iseg_colsum_wide(const void* x, int64_t ldx, int64_t rows, int64_t cols, float* out, int accumulate, int dtype, void* ws,
                                size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(x && out && rows > 0 && cols > 0 && ldx >= cols, "iseg_colsum_wide: bad arguments");
    const int chunks = wide_chunks(rows);
    const size_t need = (size_t)chunks * (size_t)cols * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_colsum_wide: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const dim3 grid((unsigned)ceil_div64(cols, 256), chunks);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((colsum_wide_partial_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)x, ldx, rows, cols, (float*)ws);
    else
        hipLaunchKernelGGL((colsum_wide_partial_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, ldx, rows, cols, (float*)ws);
    launch_reduce_rows((const float*)ws, chunks, cols, 0, 1, cols, out, nullptr, cols, 0, 1.f, accumulate, stream);
    return iseg_check_launch("iseg_colsum_wide");
}

extern "C" int iseg_relpos_bias_gather(const float* table, const int32_t* index, float* bias, int heads, int TT, hipStream_t stream) {
    ISEG_REQUIRE(table && index && bias && heads > 0 && TT > 0, "iseg_relpos_bias_gather: bad arguments");
    hipLaunchKernelGGL(relpos_gather_kernel, dim3((heads * TT + 255) / 256), dim3(256), 0, stream, table, index, bias, heads, TT);
    return iseg_check_launch("iseg_relpos_bias_gather");
}

extern "C" int iseg_relpos_bias_scatter_grad(const float* dbias, int ld, const int32_t* index, float* dtable, int entries, int heads,
                                             int T, int accumulate, hipStream_t stream) {
    ISEG_REQUIRE(dbias && index && dtable && entries > 0 && heads > 0 && T > 0 && ld >= T, "iseg_relpos_bias_scatter_grad: bad arguments");
    hipLaunchKernelGGL(relpos_scatter_kernel, dim3((entries * heads + 255) / 256), dim3(256), 0, stream, dbias, ld, index, dtable,
                       entries, heads, T, accumulate);
    return iseg_check_launch("iseg_relpos_bias_scatter_grad");
}
