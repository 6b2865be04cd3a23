// HBM-bound helpers of the heads / alternative backbones:
//   * replace_nan_or_inf          utils/op_utils.py:43-60 of the reference (FPN skip features, layers/fpn.py:52)
//   * GroupNormalization          layers/groupnorm.py:148-207 (moments over H,W,C/G per sample and group)
//   * RMSNormalization            layers/rmsnorm.py:22-29     (x * rsqrt(mean(x^2)+eps) * (1+scale), fp32 math)
//   * max / average pooling SAME  backbones/resnet_common.py:215-217 (3x3/s2 max), resnet_blocks.py:182-186 (avg shortcut)
#include "common.h"
#include "iseg_hip.h"

#include <float.h>

namespace {

// ---- block-wide deterministic sum (256 threads): wave butterflies, then the four wave totals in wave order --------------
__device__ __forceinline__ float block_sum_256(float v, float* scratch /*[4]*/) {
    v = wave_sum(v);
    __syncthreads();  // scratch may still be read from a previous call
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// ---- replace_nan_or_inf ------------------------------------------------------------------------------------------------
// order-preserving float <-> uint encoding so that min / max can use integer atomics (order independent => deterministic)
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

__global__ void minmax_init_kernel(unsigned* mm) {
    mm[0] = 0xffffffffu;  // running min (ordered encoding)
    mm[1] = 0u;           // running max
}

template <class T>
__global__ __launch_bounds__(256) void finite_minmax_kernel(const T* __restrict__ x, int64_t n, float nan_value,
                                                            unsigned* __restrict__ mm) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    auto take = [&](float v) {
        if (v != v) v = nan_value;          // replace_nan
        if (isinf(v)) v = 0.f;              // replace_inf: inf -> 0 before the global min / max
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    };
    const int64_t n8 = ((uintptr_t)x % 16 == 0) ? n / 8 : 0;   // 16-byte (bf16) / 32-byte (fp32) vector body, scalar tail
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int u = 0; u < 8; ++u) take(v[u]);
    }
    for (int64_t i = n8 * 8 + blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) take(to_f32(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    // one pair of atomics per workgroup (the grid is capped at a few thousand workgroups): per-wave atomics on two addresses
    // serialised 400 k requests at 25 M elements (379 us measured)
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6] = lo;
        shi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        hi = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
        // 4096 workgroups x 2 atomics on two addresses serialise (97 us whatever the tensor size, Swin-T + FPN): look first, and only a workgroup that
        // would move the running value sends its atomic -- after the first arrivals almost none does.  A stale look costs one needless atomic, never a
        // wrong result (min / max are order independent).
        const unsigned olo = f2ord(lo), ohi = f2ord(hi);
        if (olo < __hip_atomic_load(&mm[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mm[0], olo);
        if (ohi > __hip_atomic_load(&mm[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&mm[1], ohi);
    }
}

template <class T>
__global__ __launch_bounds__(256) void sanitize_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float nan_value,
                                                             const unsigned* __restrict__ mm) {
    const float lo = ord2f(mm[0]), hi = ord2f(mm[1]);
    auto fix = [&](float v) {
        if (v != v) v = nan_value;
        return fminf(fmaxf(v, lo), hi);     // tf.clip_by_value(x, min, max)
    };
    const int64_t n8 = (((uintptr_t)x | (uintptr_t)y) % 16 == 0) ? n / 8 : 0;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = fix(v[u]);
        store8<T>(y + i * 8, v);
    }
    for (int64_t i = n8 * 8 + blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = from_f32<T>(fix(to_f32(x[i])));
}

// gradient: tf.where(is_nan) blocks NaN positions, clip_by_value passes min <= x <= max (finite values always are)
template <class T>
__global__ __launch_bounds__(256) void sanitize_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                           int64_t n) {
    const int64_t n8 = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % 16 == 0) ? n / 8 : 0;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8], d[8];
        load8<T>(x + i * 8, v);
        load8<T>(dy + i * 8, d);
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = ((v[u] == v[u]) && !isinf(v[u])) ? d[u] : 0.f;
        store8<T>(dx + i * 8, d);
    }
    for (int64_t i = n8 * 8 + blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = to_f32(x[i]);
        const bool pass = (v == v) && !isinf(v);
        dx[i] = pass ? dy[i] : from_f32<T>(0.f);
    }
}

// ---- GroupNorm ---------------------------------------------------------------------------------------------------------
// one workgroup per (sample, group): the group's H*W x Cg slice (row stride C) is walked with lane -> channel, so a wave
// touches contiguous Cg-element runs; three passes (mean, centred variance, apply) -- the slice stays in L2.
template <class T>
__global__ __launch_bounds__(256) void groupnorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int HW,
                                                            int C, int G, float eps) {
    __shared__ float scratch[4];
    const int Cg = C / G;
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int tpr = Cg < 256 ? Cg : 256;       // threads per pixel row
    const int R = 256 / tpr;                   // pixel rows per sweep
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const bool active = tr < R;
    const T* xg = x + (int64_t)n * HW * C + g * Cg;
    T* yg = y + (int64_t)n * HW * C + g * Cg;
    const float inv_m = 1.0f / ((float)HW * (float)Cg);
    float s = 0.f;
    if (active)
        for (int p = tr; p < HW; p += R)
            for (int c = tc; c < Cg; c += tpr) s += to_f32(xg[(int64_t)p * C + c]);
    const float mean = block_sum_256(s, scratch) * inv_m;
    float q = 0.f;
    if (active)
        for (int p = tr; p < HW; p += R)
            for (int c = tc; c < Cg; c += tpr) {
                const float d = to_f32(xg[(int64_t)p * C + c]) - mean;
                q = fmaf(d, d, q);
            }
    const float var = block_sum_256(q, scratch) * inv_m;
    const float rstd = rsqrtf(var + eps);
    if (threadIdx.x == 0) {
        mean_out[blockIdx.x] = mean;
        rstd_out[blockIdx.x] = rstd;
    }
    if (active)
        for (int c = tc; c < Cg; c += tpr) {
            const float ga = gamma ? gamma[g * Cg + c] : 1.f, be = beta ? beta[g * Cg + c] : 0.f;
            const float a = rstd * ga, b = be - mean * a;   // tf.nn.batch_normalization: x*inv + (beta - mean*inv)
            for (int p = tr; p < HW; p += R) yg[(int64_t)p * C + c] = from_f32<T>(fmaf(to_f32(xg[(int64_t)p * C + c]), a, b));
        }
}

// backward of one (sample, group): s1 = sum dy*gamma, s2 = sum dy*gamma*xhat;  dx = rstd*(dy*gamma - s1/m - xhat*s2/m);
// per-sample parameter-gradient partials [n][2][C] (dgamma = sum dy*xhat, dbeta = sum dy) reduced later in sample order.
template <class T>
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, T* __restrict__ dx,
                                                            float* __restrict__ partials, int HW, int C, int G) {
    __shared__ float scratch[4];
    __shared__ float red[2][256];
    const int Cg = C / G;                      // host guarantees Cg <= 256
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int R = 256 / Cg;
    const int tc = threadIdx.x % Cg, tr = threadIdx.x / Cg;
    const bool active = tr < R;
    const int64_t base = (int64_t)n * HW * C + g * Cg;
    const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
    const float ga = gamma ? gamma[g * Cg + tc] : 1.f;
    float s1 = 0.f, s2 = 0.f, dg = 0.f, db = 0.f;
    if (active)
        for (int p = tr; p < HW; p += R) {
            const float d = to_f32(dy[base + (int64_t)p * C + tc]);
            const float xh = (to_f32(x[base + (int64_t)p * C + tc]) - mean) * rstd;
            dg = fmaf(d, xh, dg);
            db += d;
            s1 = fmaf(d, ga, s1);
            s2 = fmaf(d * ga, xh, s2);
        }
    const float S1 = block_sum_256(s1, scratch);
    const float S2 = block_sum_256(s2, scratch);
    red[0][threadIdx.x] = active ? dg : 0.f;
    red[1][threadIdx.x] = active ? db : 0.f;
    __syncthreads();
    if (threadIdx.x < Cg) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < R; ++r) {
            a += red[0][r * Cg + threadIdx.x];
            b += red[1][r * Cg + threadIdx.x];
        }
        partials[(int64_t)n * 2 * C + g * Cg + threadIdx.x] = a;
        partials[(int64_t)n * 2 * C + C + g * Cg + threadIdx.x] = b;
    }
    const float inv_m = 1.0f / ((float)HW * (float)Cg);
    const float m1 = S1 * inv_m, m2 = S2 * inv_m;
    if (active)
        for (int p = tr; p < HW; p += R) {
            const float d = to_f32(dy[base + (int64_t)p * C + tc]);
            const float xh = (to_f32(x[base + (int64_t)p * C + tc]) - mean) * rstd;
            dx[base + (int64_t)p * C + tc] = from_f32<T>(rstd * (d * ga - m1 - xh * m2));
        }
}

// ---- RMSNorm -----------------------------------------------------------------------------------------------------------
// one wave per row, lanes stride the channel axis
template <class T>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y,
                                                          float* __restrict__ rstd_out, int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = blockIdx.x * 4ll + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const T* xr = x + r * C;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = to_f32(xr[c]);
            q = fmaf(v, v, q);
        }
        q = wave_sum(q);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);   // tf.math.reciprocal(tf.sqrt(var + eps))
        if (lane == 0 && rstd_out) rstd_out[r] = rstd;
        for (int c = lane; c < C; c += 64) y[r * C + c] = from_f32<T>(to_f32(xr[c]) * rstd * (1.0f + scale[c]));
    }
}

// dx = rstd*(dy*g - xhat*mean_c(dy*g*xhat)), g = 1+scale, xhat = x*rstd;  dscale[c] = sum_rows dy*xhat.
// Each wave owns an LDS slab of C floats (lane <-> channel ownership, no atomics), slabs are added in wave order.
template <class T>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ scale, const float* __restrict__ rstd_in,
                                                          T* __restrict__ dx, float* __restrict__ partials, int64_t rows, int C) {
    extern __shared__ float slab[];  // [4][C]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* mine = slab + (size_t)wv * C;
    for (int c = lane; c < C; c += 64) mine[c] = 0.f;
    for (int64_t r = blockIdx.x * 4ll + wv; r < rows; r += (int64_t)gridDim.x * 4) {
        const float rstd = rstd_in[r];
        float s = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float d = to_f32(dy[r * C + c]), xh = to_f32(x[r * C + c]) * rstd;
            s = fmaf(d * (1.0f + scale[c]), xh, s);
            mine[c] = fmaf(d, xh, mine[c]);
        }
        s = wave_sum(s) / (float)C;
        for (int c = lane; c < C; c += 64) {
            const float d = to_f32(dy[r * C + c]), xh = to_f32(x[r * C + c]) * rstd;
            dx[r * C + c] = from_f32<T>(rstd * (d * (1.0f + scale[c]) - xh * s));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        partials[(int64_t)blockIdx.x * C + c] = (slab[c] + slab[C + c]) + (slab[2 * C + c] + slab[3 * C + c]);
}

static inline int rms_blocks(int64_t rows) {
    int64_t b = ceil_div64(rows, 4 * 8);
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- pooling, padding="SAME" -----------------------------------------------------------------------------------------
// mode 0: max (padding never wins), mode 1: average over the valid cells only (tf.nn.avg_pool2d SAME)
template <class T>
__global__ __launch_bounds__(256) void pool2d_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                                         int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo, int mode) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ow = (int)(t % Wo);
        t /= Wo;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float acc = mode == 0 ? -FLT_MAX : 0.f;
        int cnt = 0;
        for (int a = 0; a < kh; ++a) {
            const int ih = oh * sh - pt + a;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int b = 0; b < kw; ++b) {
                const int iw = ow * sw - pl + b;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float v = to_f32(x[(((int64_t)n * H + ih) * W + iw) * C + c]);
                acc = mode == 0 ? fmaxf(acc, v) : acc + v;
                ++cnt;
            }
        }
        y[i] = from_f32<T>(mode == 0 ? acc : acc / (float)cnt);
    }
}

// gather form of the gradient (deterministic, no atomics): an input cell visits every window that covers it.
// max: the window's gradient goes to its FIRST maximal cell in row-major window order (TF MaxPoolGrad).
template <class T>
__global__ __launch_bounds__(256) void pool2d_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int N,
                                                         int H, int W, int C, int kh, int kw, int sh, int sw, int pt, int pl, int Ho,
                                                         int Wo, int mode) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int iw = (int)(t % W);
        t /= W;
        const int ih = (int)(t % H);
        const int n = (int)(t / H);
        // windows oh with oh*sh - pt <= ih <= oh*sh - pt + kh - 1
        const int oh_lo = max(0, (ih + pt - kh + 1 + sh - 1) / sh), oh_hi = min(Ho - 1, (ih + pt) / sh);
        const int ow_lo = max(0, (iw + pl - kw + 1 + sw - 1) / sw), ow_hi = min(Wo - 1, (iw + pl) / sw);
        const float mine = mode == 0 ? to_f32(x[i]) : 0.f;
        float g = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const float d = to_f32(dy[(((int64_t)n * Ho + oh) * Wo + ow) * C + c]);
                if (mode == 1) {
                    const int h0 = max(0, oh * sh - pt), h1 = min(H, oh * sh - pt + kh);
                    const int w0 = max(0, ow * sw - pl), w1 = min(W, ow * sw - pl + kw);
                    g += d / (float)((h1 - h0) * (w1 - w0));
                } else {
                    bool winner = true;   // no earlier cell >= mine, no later cell > mine
                    for (int a = 0; a < kh && winner; ++a) {
                        const int yh = oh * sh - pt + a;
                        if ((unsigned)yh >= (unsigned)H) continue;
                        for (int b = 0; b < kw; ++b) {
                            const int yw = ow * sw - pl + b;
                            if ((unsigned)yw >= (unsigned)W) continue;
                            const float v = to_f32(x[(((int64_t)n * H + yh) * W + yw) * C + c]);
                            const bool earlier = yh < ih || (yh == ih && yw < iw);
                            if (earlier ? v >= mine : v > mine) {
                                winner = false;
                                break;
                            }
                        }
                    }
                    if (winner) g += d;
                }
            }
        dx[i] = from_f32<T>(g);
    }
}

// Max-pool gradient in two vector passes (C % 8 == 0, one byte of workspace per output element): pass 1 records, per window and
// channel, the row-major index of its FIRST maximal cell (TF MaxPoolGrad's tie rule); pass 2 lets every input cell look up the
// <= ceil(kh/sh)*ceil(kw/sw) windows that cover it and take their gradient where the recorded index is its own.  8 channels per lane,
// 16-byte loads.  The one-pass kernel above re-derives the winner of every covering window per input element from scalar loads
// (36 two-byte loads per element for the 3x3/s2 ResNet stem pool: 388 us on 16x128x128x128).
template <class T>
__global__ __launch_bounds__(256) void pool_argmax_kernel(const T* __restrict__ x, unsigned char* __restrict__ idx, int N, int H, int W,
                                                          int C, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo) {
    const int C8 = C / 8;
    const int64_t total = (int64_t)N * Ho * Wo * C8;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        int64_t t = i / C8;
        const int ow = (int)(t % Wo);
        t /= Wo;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float best[8];
        unsigned int bi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            best[u] = -FLT_MAX;
            bi[u] = 255u;
        }
        for (int a = 0; a < kh; ++a) {
            const int yh = oh * sh - pt + a;
            if ((unsigned)yh >= (unsigned)H) continue;
            for (int b = 0; b < kw; ++b) {
                const int yw = ow * sw - pl + b;
                if ((unsigned)yw >= (unsigned)W) continue;
                float v[8];
                load8<T>(x + (((int64_t)n * H + yh) * W + yw) * C + c8 * 8, v);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (bi[u] == 255u || v[u] > best[u]) {      // strict: the first maximal cell keeps the window
                        best[u] = v[u];
                        bi[u] = (unsigned)(a * kw + b);
                    }
            }
        }
        uint2 packed;
        packed.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        packed.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
        *reinterpret_cast<uint2*>(idx + i * 8) = packed;
    }
}

template <class T>
__global__ __launch_bounds__(256) void pool_max_bwd_idx_kernel(const unsigned char* __restrict__ idx, const T* __restrict__ dy,
                                                               T* __restrict__ dx, int N, int H, int W, int C, int kh, int kw, int sh,
                                                               int sw, int pt, int pl, int Ho, int Wo) {
    const int C8 = C / 8;
    const int64_t total = (int64_t)N * H * W * C8;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        int64_t t = i / C8;
        const int iw = (int)(t % W);
        t /= W;
        const int ih = (int)(t % H);
        const int n = (int)(t / H);
        const int oh_lo = max(0, (ih + pt - kh + 1 + sh - 1) / sh), oh_hi = min(Ho - 1, (ih + pt) / sh);
        const int ow_lo = max(0, (iw + pl - kw + 1 + sw - 1) / sw), ow_hi = min(Wo - 1, (iw + pl) / sw);
        float g[8] = {};
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const unsigned cell = (unsigned)((ih - (oh * sh - pt)) * kw + (iw - (ow * sw - pl)));
                const int64_t o = (((int64_t)n * Ho + oh) * Wo + ow) * C8 + c8;
                const uint2 packed = *reinterpret_cast<const uint2*>(idx + o * 8);
                float d[8];
                load8<T>(dy + o * 8, d);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned w = ((u < 4 ? packed.x : packed.y) >> (8 * (u & 3))) & 255u;
                    if (w == cell) g[u] += d[u];
                }
            }
        store8<T>(dx + i * 8, g);
    }
}

// residual join of the ResNet bottleneck: relu(a + b)   (backbones/resnet_blocks.py:106-107,202-203)
template <class T>
__global__ __launch_bounds__(256) void add_relu_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n8) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float u[8], v[8];
        load8<T>(a + i * 8, u);
        load8<T>(b + i * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = fmaxf(u[k] + v[k], 0.f);
        store8<T>(y + i * 8, u);
    }
}

static inline unsigned ew_blocks(int64_t n) {
    int64_t b = ceil_div64(n, 256);
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int iseg_replace_nan_or_inf(const void* x, void* y, int64_t n, float nan_value, int dtype, void* ws, size_t ws_bytes,
                                       hipStream_t stream) {
    ISEG_REQUIRE(x && y && n > 0, "iseg_replace_nan_or_inf: bad arguments");
    if (!ws || ws_bytes < 8) {
        iseg_set_error("iseg_replace_nan_or_inf: needs 8 workspace bytes, got %zu", ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    unsigned* mm = (unsigned*)ws;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, stream, mm);
    // the min / max pass: at most four workgroups per CU (each ends in up to two atomics on the same two words), eight elements per lane and trip
    int64_t rb = ceil_div64(ceil_div64(n, 8), 256);
    if (rb > 1024) rb = 1024;
    const unsigned red_blocks = (unsigned)rb;
    if (dtype == ISEG_BF16) {
        hipLaunchKernelGGL((finite_minmax_kernel<bf16_t>), dim3(red_blocks), dim3(256), 0, stream, (const bf16_t*)x, n, nan_value, mm);
        hipLaunchKernelGGL((sanitize_apply_kernel<bf16_t>), dim3(ew_blocks(n)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n,
                           nan_value, mm);
    } else {
        hipLaunchKernelGGL((finite_minmax_kernel<float>), dim3(red_blocks), dim3(256), 0, stream, (const float*)x, n, nan_value, mm);
        hipLaunchKernelGGL((sanitize_apply_kernel<float>), dim3(ew_blocks(n)), dim3(256), 0, stream, (const float*)x, (float*)y, n,
                           nan_value, mm);
    }
    return iseg_check_launch("iseg_replace_nan_or_inf");
}

extern "C" int iseg_replace_nan_or_inf_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && dy && dx && n > 0, "iseg_replace_nan_or_inf_bwd: bad arguments");
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((sanitize_bwd_kernel<bf16_t>), dim3(ew_blocks(n)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy,
                           (bf16_t*)dx, n);
    else
        hipLaunchKernelGGL((sanitize_bwd_kernel<float>), dim3(ew_blocks(n)), dim3(256), 0, stream, (const float*)x, (const float*)dy,
                           (float*)dx, n);
    return iseg_check_launch("iseg_replace_nan_or_inf_bwd");
}

extern "C" int iseg_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int N,
                                  int HW, int C, int G, float eps, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && y && mean && rstd, "iseg_groupnorm_fwd: null pointer");
    ISEG_REQUIRE(N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "iseg_groupnorm_fwd: C=%d is not a multiple of groups=%d", C, G);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((groupnorm_fwd_kernel<bf16_t>), dim3(N * G), dim3(256), 0, stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y,
                           mean, rstd, HW, C, G, eps);
    else
        hipLaunchKernelGGL((groupnorm_fwd_kernel<float>), dim3(N * G), dim3(256), 0, stream, (const float*)x, gamma, beta, (float*)y,
                           mean, rstd, HW, C, G, eps);
    return iseg_check_launch("iseg_groupnorm_fwd");
}

extern "C" size_t iseg_groupnorm_bwd_workspace_bytes(int N, int C) { return (size_t)N * 2 * C * sizeof(float); }

extern "C" int iseg_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                                  float* dgamma, float* dbeta, int accumulate_param_grads, int N, int HW, int C, int G, int dtype,
                                  void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(dy && x && mean && rstd && dx, "iseg_groupnorm_bwd: null pointer");
    ISEG_REQUIRE(N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "iseg_groupnorm_bwd: C=%d is not a multiple of groups=%d", C, G);
    ISEG_REQUIRE(C / G <= 256, "iseg_groupnorm_bwd: %d channels per group (max 256)", C / G);
    const size_t need = iseg_groupnorm_bwd_workspace_bytes(N, C);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_groupnorm_bwd: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((groupnorm_bwd_kernel<bf16_t>), dim3(N * G), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, gamma,
                           mean, rstd, (bf16_t*)dx, (float*)ws, HW, C, G);
    else
        hipLaunchKernelGGL((groupnorm_bwd_kernel<float>), dim3(N * G), dim3(256), 0, stream, (const float*)dy, (const float*)x, gamma,
                           mean, rstd, (float*)dx, (float*)ws, HW, C, G);
    if (dgamma || dbeta) {
        if (dgamma) launch_reduce_rows((const float*)ws, N, 2 * C, 0, 1, C, dgamma, nullptr, C, 0, 1.f, accumulate_param_grads, stream);
        if (dbeta) launch_reduce_rows((const float*)ws + C, N, 2 * C, 0, 1, C, dbeta, nullptr, C, 0, 1.f, accumulate_param_grads, stream);
    }
    return iseg_check_launch("iseg_groupnorm_bwd");
}

extern "C" int iseg_rmsnorm_fwd(const void* x, const float* scale, void* y, float* rstd, int64_t rows, int C, float eps, int dtype,
                                hipStream_t stream) {
    ISEG_REQUIRE(x && scale && y && rows > 0 && C > 0, "iseg_rmsnorm_fwd: bad arguments");
    const int blocks = rms_blocks(rows);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, scale, (bf16_t*)y, rstd,
                           rows, C, eps);
    else
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)x, scale, (float*)y, rstd, rows,
                           C, eps);
    return iseg_check_launch("iseg_rmsnorm_fwd");
}

extern "C" size_t iseg_rmsnorm_bwd_workspace_bytes(int64_t rows, int C) { return (size_t)rms_blocks(rows) * C * sizeof(float); }

extern "C" int iseg_rmsnorm_bwd(const void* dy, const void* x, const float* scale, const float* rstd, void* dx, float* dscale,
                                int accumulate_param_grads, int64_t rows, int C, int dtype, void* ws, size_t ws_bytes,
                                hipStream_t stream) {
    ISEG_REQUIRE(dy && x && scale && rstd && dx && dscale && rows > 0 && C > 0, "iseg_rmsnorm_bwd: bad arguments");
    ISEG_REQUIRE(C <= 8192, "iseg_rmsnorm_bwd: C=%d too wide (max 8192)", C);
    const int blocks = rms_blocks(rows);
    const size_t need = (size_t)blocks * C * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_rmsnorm_bwd: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const size_t lds = (size_t)4 * C * sizeof(float);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), lds, stream, (const bf16_t*)dy, (const bf16_t*)x, scale,
                           rstd, (bf16_t*)dx, (float*)ws, rows, C);
    else
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<float>), dim3(blocks), dim3(256), lds, stream, (const float*)dy, (const float*)x, scale,
                           rstd, (float*)dx, (float*)ws, rows, C);
    launch_reduce_rows((const float*)ws, blocks, C, 0, 1, C, dscale, nullptr, C, 0, 1.f, accumulate_param_grads, stream);
    return iseg_check_launch("iseg_rmsnorm_bwd");
}

extern "C" int iseg_pool2d_fwd(const void* x, void* y, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int pad_t, int pad_l,
                               int Ho, int Wo, int mode, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "iseg_pool2d_fwd: bad arguments");
    ISEG_REQUIRE(mode == 0 || mode == 1, "iseg_pool2d_fwd: mode %d (0 = max, 1 = avg)", mode);
    ISEG_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && pad_t >= 0 && pad_l >= 0 && pad_t < kh && pad_l < kw,
                 "iseg_pool2d_fwd: bad window geometry");
    ISEG_REQUIRE((Ho - 1) * sh - pad_t < H && (Wo - 1) * sw - pad_l < W, "iseg_pool2d_fwd: output window outside the input");
    const int64_t total = (int64_t)N * Ho * Wo * C;
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((pool2d_fwd_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, N, H, W,
                           C, kh, kw, sh, sw, pad_t, pad_l, Ho, Wo, mode);
    else
        hipLaunchKernelGGL((pool2d_fwd_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, stream, (const float*)x, (float*)y, N, H, W, C,
                           kh, kw, sh, sw, pad_t, pad_l, Ho, Wo, mode);
    return iseg_check_launch("iseg_pool2d_fwd");
}

extern "C" size_t iseg_pool2d_bwd_workspace_bytes(int N, int Ho, int Wo, int C, int mode) {
    return (mode == 0 && C % 8 == 0) ? (size_t)N * Ho * Wo * C : 0;      // one winner index per window and channel
}

extern "C" int iseg_pool2d_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                               int pad_t, int pad_l, int Ho, int Wo, int mode, int dtype, void* ws, size_t ws_bytes,
                               hipStream_t stream) {
    ISEG_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "iseg_pool2d_bwd: bad arguments");
    ISEG_REQUIRE(mode == 0 || mode == 1, "iseg_pool2d_bwd: mode %d (0 = max, 1 = avg)", mode);
    ISEG_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && pad_t >= 0 && pad_l >= 0 && pad_t < kh && pad_l < kw,
                 "iseg_pool2d_bwd: bad window geometry");
    const int64_t total = (int64_t)N * H * W * C;
    const size_t need = iseg_pool2d_bwd_workspace_bytes(N, Ho, Wo, C, mode);
    const bool aligned = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0 && ((uintptr_t)ws & 7) == 0;
    if (need > 0 && kh * kw < 255 && aligned) {
        if (!ws || ws_bytes < need) {
            iseg_set_error("iseg_pool2d_bwd: needs %zu workspace bytes, got %zu", need, ws_bytes);
            return ISEG_ERR_WORKSPACE;
        }
        unsigned char* idx = (unsigned char*)ws;
        const int64_t n_out = (int64_t)N * Ho * Wo * (C / 8), n_in = total / 8;
        if (dtype == ISEG_BF16) {
            hipLaunchKernelGGL((pool_argmax_kernel<bf16_t>), dim3(ew_blocks(n_out)), dim3(256), 0, stream, (const bf16_t*)x, idx, N, H, W, C,
                               kh, kw, sh, sw, pad_t, pad_l, Ho, Wo);
            hipLaunchKernelGGL((pool_max_bwd_idx_kernel<bf16_t>), dim3(ew_blocks(n_in)), dim3(256), 0, stream, idx, (const bf16_t*)dy,
                               (bf16_t*)dx, N, H, W, C, kh, kw, sh, sw, pad_t, pad_l, Ho, Wo);
        } else {
            hipLaunchKernelGGL((pool_argmax_kernel<float>), dim3(ew_blocks(n_out)), dim3(256), 0, stream, (const float*)x, idx, N, H, W, C,
                               kh, kw, sh, sw, pad_t, pad_l, Ho, Wo);
            hipLaunchKernelGGL((pool_max_bwd_idx_kernel<float>), dim3(ew_blocks(n_in)), dim3(256), 0, stream, idx, (const float*)dy,
                               (float*)dx, N, H, W, C, kh, kw, sh, sw, pad_t, pad_l, Ho, Wo);
        }
        return iseg_check_launch("iseg_pool2d_bwd");
    }
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((pool2d_bwd_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy,
                           (bf16_t*)dx, N, H, W, C, kh, kw, sh, sw, pad_t, pad_l, Ho, Wo, mode);
    else
        hipLaunchKernelGGL((pool2d_bwd_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, stream, (const float*)x, (const float*)dy,
                           (float*)dx, N, H, W, C, kh, kw, sh, sw, pad_t, pad_l, Ho, Wo, mode);
    return iseg_check_launch("iseg_pool2d_bwd");
}

extern "C" int iseg_add_relu(const void* a, const void* b, void* y, int64_t n, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(a && b && y && n > 0 && n % 8 == 0, "iseg_add_relu: n=%lld must be a positive multiple of 8", (long long)n);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((add_relu_kernel<bf16_t>), dim3(ew_blocks(n / 8)), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b,
                           (bf16_t*)y, n / 8);
    else
        hipLaunchKernelGGL((add_relu_kernel<float>), dim3(ew_blocks(n / 8)), dim3(256), 0, stream, (const float*)a, (const float*)b,
                           (float*)y, n / 8);
    return iseg_check_launch("iseg_add_relu");
}
