// tf.image.resize(method=bilinear / nearest, half-pixel centres, no antialias) on NHWC tensors, as called by
// utils/common.py:107-134 resize_image (logits -> input size in layers/core_model_ext.py:217-226; FPN top-down
// path layers/fpn.py:40-61; multi-scale inference core_model.py:170-229).
//   src = (dst + 0.5) * in/out - 0.5 ; lo = max(floor(src),0) ; hi = min(ceil(src), in-1) ; t = src - floor(src)
//   out = top + (bottom - top)*ty with top = tl + (tr - tl)*tx            (lerp order kept: it fixes rounding)
// HBM-bound.  Forward: one workgroup per output row (n, oy); the two source rows are L1-resident, the output row
// is written with 16-B lanes and 32-bit index math.  Backward: the exact transpose in gather form, separated into an
// X pass (one workgroup per gradient row, staged once through LDS with coalesced 16-B loads) and a Y pass, so that no
// atomics are needed and the result is deterministic.
#include "common.h"
#include "iseg_hip.h"

namespace {

struct Lerp {
    int lo, hi;
    float t;
};

// AC: tf.compat.v1.image.resize(..., align_corners=True) -- src = dst * (in-1)/(out-1) (`scale` is that ratio); else TF2's half-pixel centres
template <bool AC = false>
__device__ __forceinline__ Lerp lerp_of(int dst, float scale, int in_size) {
    const float src = AC ? (float)dst * scale : ((float)dst + 0.5f) * scale - 0.5f;
    const float f = floorf(src);
    Lerp l;
    l.lo = max((int)f, 0);
    l.hi = min((int)ceilf(src), in_size - 1);
    l.t = src - f;
    return l;
}

template <class TO> __device__ __forceinline__ void store4(TO* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v) {
    bf16x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = (bf16_t)v[u];
    *reinterpret_cast<bf16x4*>(p) = o;
}

// grid.x = N*Ho output rows (grid-stride), 256 threads walk the Wo*C elements of the row four at a time
template <class TI, class TO, bool AC = false>
__global__ __launch_bounds__(256) void resize_bilinear_fwd_kernel(const TI* __restrict__ x, TO* __restrict__ y, int N, int Hi, int Wi,
                                                                  int Ho, int Wo, int C, float sy, float sx) {
    const int rowlen = Wo * C;
    const bool vec = (rowlen % 4 == 0);
    for (int row = blockIdx.x; row < N * Ho; row += gridDim.x) {
        const int n = row / Ho, oy = row - n * Ho;
        const Lerp ly = lerp_of<AC>(oy, sy, Hi);
        const TI* top = x + ((int64_t)n * Hi + ly.lo) * Wi * C;
        const TI* bot = x + ((int64_t)n * Hi + ly.hi) * Wi * C;
        TO* out = y + (int64_t)row * rowlen;
        for (int e0 = threadIdx.x * 4; e0 < rowlen; e0 += 256 * 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u;
                v[u] = 0.f;
                if (e < rowlen) {
                    const int ox = e / C, c = e - ox * C;
                    const Lerp lx = lerp_of<AC>(ox, sx, Wi);
                    const float tl = to_f32(top[lx.lo * C + c]), tr = to_f32(top[lx.hi * C + c]);
                    const float bl = to_f32(bot[lx.lo * C + c]), br = to_f32(bot[lx.hi * C + c]);
                    const float tp = tl + (tr - tl) * lx.t;
                    const float bt = bl + (br - bl) * lx.t;
                    v[u] = tp + (bt - tp) * ly.t;
                }
            }
            if (vec) store4<TO>(out + e0, v);
            else
                for (int u = 0; u < 4 && e0 + u < rowlen; ++u) out[e0 + u] = from_f32<TO>(v[u]);
        }
    }
}

// Large up-sampling of small maps (logits 16x16x21 -> 512x512x21 at cfg2: 88 MB written): the generic kernel above spends its
// time in per-element integer division by C, lerp_of and four dependent global gathers (170 us measured, 0.5 TB/s).  Here the two
// source rows (fp32) and the per-column lerp table sit in LDS, (ox, c) advance incrementally, and the arithmetic keeps TF's order.
template <class TI, class TO>
__global__ __launch_bounds__(256) void resize_bilinear_fwd_lds_kernel(const TI* __restrict__ x, TO* __restrict__ y, int N, int Hi, int Wi,
                                                                      int Ho, int Wo, int C, float sy, float sx) {
    extern __shared__ __attribute__((aligned(16))) float rsm[];
    const int src_len = Wi * C;
    float* top = rsm;                                   // [Wi*C]
    float* bot = rsm + src_len;                         // [Wi*C]
    int* xlo = reinterpret_cast<int*>(rsm + 2 * src_len);   // [Wo] element offsets lo*C
    int* xhi = xlo + Wo;                                // [Wo] element offsets hi*C
    float* xt = reinterpret_cast<float*>(xhi + Wo);     // [Wo]
    for (int ox = threadIdx.x; ox < Wo; ox += 256) {
        const Lerp lx = lerp_of(ox, sx, Wi);
        xlo[ox] = lx.lo * C;
        xhi[ox] = lx.hi * C;
        xt[ox] = lx.t;
    }
    const int rowlen = Wo * C;
    const bool vec = (rowlen % 4 == 0);
    const int dq = 1024 / C, dr = 1024 % C;
    for (int row = blockIdx.x; row < N * Ho; row += gridDim.x) {
        const int n = row / Ho, oy = row - n * Ho;
        const Lerp ly = lerp_of(oy, sy, Hi);
        const TI* gt = x + ((int64_t)n * Hi + ly.lo) * src_len;
        const TI* gb = x + ((int64_t)n * Hi + ly.hi) * src_len;
        __syncthreads();   // previous row consumed (and the lerp table written, first time round)
        for (int i = threadIdx.x; i < src_len; i += 256) {
            top[i] = to_f32(gt[i]);
            bot[i] = to_f32(gb[i]);
        }
        __syncthreads();
        TO* out = y + (int64_t)row * rowlen;
        int e0 = threadIdx.x * 4;
        int ox = e0 / C, c = e0 - ox * C;
        for (; e0 < rowlen; e0 += 1024) {
            float v[4];
            int oxx = ox, cc = c;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = 0.f;
                if (e0 + u < rowlen) {
                    const int lo = xlo[oxx] + cc, hi = xhi[oxx] + cc;
                    const float t = xt[oxx];
                    const float tl = top[lo], tr = top[hi], bl = bot[lo], br = bot[hi];
                    const float tp = tl + (tr - tl) * t;
                    const float bt = bl + (br - bl) * t;
                    v[u] = tp + (bt - tp) * ly.t;
                }
                if (++cc >= C) {
                    cc = 0;
                    ++oxx;
                }
            }
            if (vec) store4<TO>(out + e0, v);
            else
                for (int u = 0; u < 4 && e0 + u < rowlen; ++u) out[e0 + u] = from_f32<TO>(v[u]);
            ox += dq;
            c += dr;
            if (c >= C) {
                c -= C;
                ++ox;
            }
        }
    }
}

// transposed interpolation weights of forward destination d onto source j
template <bool AC = false>
__device__ __forceinline__ float bwd_weight(int d, int j, float scale, int J) {
    const Lerp l = lerp_of<AC>(d, scale, J);
    float w = 0.f;
    if (l.lo == j) w += 1.f - l.t;
    if (l.hi == j) w += l.t;
    return w;
}

template <bool AC = false>
__device__ __forceinline__ void bwd_range(int j, int J, int Dn, float inv, int& d0, int& d1) {
    // destinations whose lo or hi can equal j: src in (j-1, j+1)  ->  dst in ((j-0.5)*inv-0.5 , (j+1.5)*inv-0.5)
    // (aligned corners: src = dst*scale, dst in ((j-1)*inv, (j+1)*inv); a degenerate axis -- one output or one input -- scans everything)
    if (AC) {
        if (!(inv < 1e30f)) {
            d0 = 0;
            d1 = Dn - 1;
            return;
        }
        d0 = (int)floorf(((float)j - 1.f) * inv) - 1;
        d1 = (int)ceilf(((float)j + 1.f) * inv) + 1;
    } else {
        d0 = (int)floorf(((float)j - 0.5f) * inv - 0.5f) - 1;
        d1 = (int)ceilf(((float)j + 1.5f) * inv - 0.5f) + 1;
    }
    if (j == 0) d0 = 0;           // clamped sources: everything below maps to lo = hi = 0
    if (j == J - 1) d1 = Dn - 1;  // and everything above to J-1
    d0 = max(d0, 0);
    d1 = min(d1, Dn - 1);
}

// X pass with the gradient row staged in LDS: grid.x = N*Ho rows; tmp[row][ix][c] = sum_ox w(ox -> ix) dy[row][ox][c]
template <class TI>
__global__ __launch_bounds__(256) void resize_bwd_x_lds_kernel(const TI* __restrict__ dy, float* __restrict__ tmp, int rows, int Wo,
                                                               int Wi, int C, float sx) {
    extern __shared__ __attribute__((aligned(16))) float srow[];  // [Wo*C], then the destination table lo[Wo], hi[Wo], t[Wo]
    const int rowlen = Wo * C;
    const float inv = 1.0f / sx;
    constexpr int V = 16 / sizeof(TI);
    int* xlo = reinterpret_cast<int*>(srow + rowlen);
    int* xhi = xlo + Wo;
    float* xt = reinterpret_cast<float*>(xhi + Wo);
    for (int d = threadIdx.x; d < Wo; d += 256) {
        const Lerp l = lerp_of(d, sx, Wi);
        xlo[d] = l.lo;
        xhi[d] = l.hi;
        xt[d] = l.t;
    }
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const TI* src = dy + (int64_t)row * rowlen;
        __syncthreads();
        if (rowlen % V == 0 && ((int64_t)row * rowlen) % V == 0) {
            constexpr int NB = 4;      // loads issued together per lane (one per trip left ~11 dependent round trips per row)
            for (int base = threadIdx.x; base < rowlen / V; base += 256 * NB) {
                float v[NB][V];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int i = base + q * 256;
                    if (i < rowlen / V) Vec16<TI>::load(src + i * V, v[q]);
                }
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int i = base + q * 256;
                    if (i < rowlen / V) {
#pragma unroll
                        for (int u = 0; u < V; ++u) srow[i * V + u] = v[q][u];
                    }
                }
            }
        } else {
            for (int i = threadIdx.x; i < rowlen; i += 256) srow[i] = to_f32(src[i]);
        }
        __syncthreads();
        // four lanes share one output (ix, c): each walks every fourth destination of the range with the (lo, hi, t) table
        // from LDS, partial sums meet through two shuffles (same fixed order every run)
        for (int o0 = 0; o0 < Wi * C; o0 += 64) {
            const int o = o0 + (threadIdx.x >> 2), part = threadIdx.x & 3;
            float acc = 0.f;
            if (o < Wi * C) {
                const int ix = o / C, c = o - ix * C;
                int d0, d1;
                bwd_range(ix, Wi, Wo, inv, d0, d1);
                // four destinations per trip, their LDS reads issued together (the loop is bound by LDS latency, not by bandwidth)
                int d = d0 + part;
                for (; d + 12 <= d1; d += 16) {
                    float t[4], v[4];
                    int lo[4], hi[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        t[q] = xt[d + 4 * q];
                        v[q] = srow[(d + 4 * q) * C + c];
                        lo[q] = xlo[d + 4 * q];
                        hi[q] = xhi[d + 4 * q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (lo[q] == ix) acc = fmaf(1.f - t[q], v[q], acc);
                        if (hi[q] == ix) acc = fmaf(t[q], v[q], acc);
                    }
                }
                for (; d <= d1; d += 4) {
                    const float t = xt[d], v = srow[d * C + c];
                    if (xlo[d] == ix) acc = fmaf(1.f - t, v, acc);
                    if (xhi[d] == ix) acc = fmaf(t, v, acc);
                }
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (part == 0 && o < Wi * C) tmp[(int64_t)row * Wi * C + o] = acc;
        }
    }
}

// one axis of the transposed interpolation, generic gather: out[o, j, q] = sum_{d in D(j)} w(d -> j) * in[o, d, q]
//   outer o (size O), reduced axis d (size Dn, "destination" of the forward), kept axis j (size J, forward source),
//   inner q (size Q, contiguous).   scale = J / Dn (forward in/out ratio along this axis)
template <class TI, class TO, bool AC = false>
__global__ void resize_bwd_axis_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t O, int Dn, int J, int64_t Q,
                                       float scale, const TO* __restrict__ add) {
    const int64_t total = O * J * Q;
    const float inv = scale > 0.f ? 1.0f / scale : 3.0e38f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i % Q;
        const int j = (int)((i / Q) % J);
        const int64_t o = i / (Q * J);
        int d0, d1;
        bwd_range<AC>(j, J, Dn, inv, d0, d1);
        float acc = 0.f;
        const TI* p = in + o * Dn * Q + q;
        for (int d = d0; d <= d1; ++d) {
            const float w = bwd_weight<AC>(d, j, scale, J);
            if (w != 0.f) acc += w * to_f32(p[(int64_t)d * Q]);
        }
        if (add) acc += to_f32(add[i]);
        out[i] = from_f32<TO>(acc);
    }
}

// 8-element (channel) vector forms of the two generic kernels for C % 8 == 0 (FPN x2 up-sampling of 768-channel maps): index
// math and interpolation weights once per 8 channels, 16-byte (bf16) / 32-byte (fp32) accesses.
template <class TI, class TO>
__global__ __launch_bounds__(256) void resize_bilinear_fwd_vec_kernel(const TI* __restrict__ x, TO* __restrict__ y, int N, int Hi, int Wi,
                                                                      int Ho, int Wo, int C, float sy, float sx) {
    const int C8 = C / 8;
    const int64_t total = (int64_t)N * Ho * Wo * C8;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        int64_t r = i / C8;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const Lerp ly = lerp_of(oy, sy, Hi), lx = lerp_of(ox, sx, Wi);
        const TI* top = x + ((int64_t)n * Hi + ly.lo) * Wi * C + c;
        const TI* bot = x + ((int64_t)n * Hi + ly.hi) * Wi * C + c;
        float tl[8], tr[8], bl[8], br[8], v[8];
        load8<TI>(top + (int64_t)lx.lo * C, tl);
        load8<TI>(top + (int64_t)lx.hi * C, tr);
        load8<TI>(bot + (int64_t)lx.lo * C, bl);
        load8<TI>(bot + (int64_t)lx.hi * C, br);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float tp = tl[u] + (tr[u] - tl[u]) * lx.t;
            const float bt = bl[u] + (br[u] - bl[u]) * lx.t;
            v[u] = tp + (bt - tp) * ly.t;
        }
        store8<TO>(y + i * 8, v);
    }
}

// out = relu((z - mean) * rstd * gamma + beta) + bilinear_up(x): one level of the FPN top-down pathway (layers/fpn.py:46-57: ConvNormAct's
// BatchNorm + ReLU, resize_image of the running map, the sum) in one pass -- neither the normalised map nor the up-sampled one is written
// (3 x 403 MB each way at the stride-4 level of Swin-T + FPN).  z, out [N, Ho, Wo, C]; x [N, Hi, Wi, C]; C % 8 == 0.
template <class T>
__global__ __launch_bounds__(256) void bn_relu_upsample_add_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const T* __restrict__ x, T* __restrict__ out,
                                                                   int N, int Hi, int Wi, int Ho, int Wo, int C, float sy, float sx, int tpc) {
    // thread (tc, tr) of a tpc x (blockDim / tpc) workgroup owns channel chunk(s) tc, tc + tpc, ... and the pixels blockIdx * rpi + tr,
    // + gridDim * rpi, ...: the four per-channel vectors are loaded once per chunk, not once per 16 bytes of z (they were 60 % of the
    // kernel's L1 traffic: 308 us for 906 MB at the stride-4 level of Swin-T + FPN)
    const int C8 = C / 8, rpi = blockDim.x / tpc;
    const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
    const unsigned npix = (unsigned)N * Ho * Wo;
    const unsigned pstep = gridDim.x * (unsigned)rpi;
    for (int cc = tc; cc < C8; cc += tpc) {
        const int c = cc * 8;
        float m[8], s[8], g[8], b[8];
        load8<float>(mean + c, m);
        load8<float>(rstd + c, s);
        load8<float>(gamma + c, g);
        load8<float>(beta + c, b);
        for (unsigned p = blockIdx.x * (unsigned)rpi + tr; p < npix; p += pstep) {
            const int ox = (int)(p % (unsigned)Wo);
            const unsigned r = p / (unsigned)Wo;
            const int oy = (int)(r % (unsigned)Ho);
            const int n = (int)(r / (unsigned)Ho);
            const Lerp ly = lerp_of(oy, sy, Hi), lx = lerp_of(ox, sx, Wi);
            const T* top = x + ((int64_t)n * Hi + ly.lo) * Wi * C + c;
            const T* bot = x + ((int64_t)n * Hi + ly.hi) * Wi * C + c;
            float tl[8], tr_[8], bl[8], br[8], v[8];
            load8<T>(z + (int64_t)p * C + c, v);
            load8<T>(top + (int64_t)lx.lo * C, tl);
            load8<T>(top + (int64_t)lx.hi * C, tr_);
            load8<T>(bot + (int64_t)lx.lo * C, bl);
            load8<T>(bot + (int64_t)lx.hi * C, br);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float tp = tl[u] + (tr_[u] - tl[u]) * lx.t;
                const float bt = bl[u] + (br[u] - bl[u]) * lx.t;
                const float y = fmaxf((v[u] - m[u]) * s[u] * g[u] + b[u], 0.f);
                v[u] = y + (tp + (bt - tp) * ly.t);
            }
            store8<T>(out + (int64_t)p * C + c, v);
        }
    }
}

// Transposed interpolation of an exact x2 up-sampling (Ho = 2 Hi, Wo = 2 Wi: every FPN level at even sizes) in ONE pass: source pixel
// (jy, jx) collects the 4 x 4 destinations 2j - 1 .. 2j + 2 of each axis with the weights the two-pass form derives (bwd_weight), all sixteen
// loads issued unconditionally from clamped addresses (no fp32 intermediate of N Ho Wi C: 167 + 114 us -> one pass at the stride-4 level).
template <class TI, class TO>
__global__ __launch_bounds__(256) void resize_bwd_x2_vec_kernel(const TI* __restrict__ dy, TO* __restrict__ dx, int N, int Hi, int Wi, int C,
                                                                const TO* __restrict__ add) {
    const unsigned C8 = C / 8;
    const unsigned total = (unsigned)N * Hi * Wi * C8;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int c = (int)(i % C8) * 8;
        unsigned r = i / C8;
        const int jx = (int)(r % (unsigned)Wi);
        r /= (unsigned)Wi;
        const int jy = (int)(r % (unsigned)Hi);
        const int n = (int)(r / (unsigned)Hi);
        float wy[4], wx[4];
        int yy[4], xx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = 2 * jy - 1 + k, e = 2 * jx - 1 + k;
            const bool vy = d >= 0 && d < Ho, vx = e >= 0 && e < Wo;
            yy[k] = vy ? d : 0;
            xx[k] = vx ? e : 0;
            wy[k] = vy ? bwd_weight(d, jy, 0.5f, Hi) : 0.f;
            wx[k] = vx ? bwd_weight(e, jx, 0.5f, Wi) : 0.f;
        }
        float v[16][8];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) load8<TI>(dy + (((int64_t)n * Ho + yy[ky]) * Wo + xx[kx]) * C + c, v[4 * ky + kx]);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            float row[8];      // the x pass of this destination row first, then the y weight: the two-pass form's order
#pragma unroll
            for (int u = 0; u < 8; ++u) row[u] = 0.f;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                for (int u = 0; u < 8; ++u) row[u] += wx[kx] * v[4 * ky + kx][u];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += wy[ky] * row[u];
        }
        if (add) {
            float a[8];
            load8<TO>(add + (int64_t)i * 8, a);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += a[u];
        }
        store8<TO>(dx + (int64_t)i * 8, acc);
    }
}

template <class TI, class TO>
__global__ __launch_bounds__(256) void resize_bwd_axis_vec_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t O, int Dn, int J,
                                                                  int64_t Q, float scale, const TO* __restrict__ add) {
    const int64_t Q8 = Q / 8;
    const int64_t total = O * J * Q8;
    const float inv = 1.0f / scale;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t q = (i % Q8) * 8;
        const int j = (int)((i / Q8) % J);
        const int64_t o = i / (Q8 * J);
        int d0, d1;
        bwd_range(j, J, Dn, inv, d0, d1);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
        const TI* p = in + o * Dn * Q + q;
        for (int d = d0; d <= d1; ++d) {
            const float w = bwd_weight(d, j, scale, J);
            if (w != 0.f) {
                float v[8];
                load8<TI>(p + (int64_t)d * Q, v);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += w * v[u];
            }
        }
        const int64_t e = (o * J + j) * Q + q;
        if (add) {
            float a[8];
            load8<TO>(add + e, a);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += a[u];
        }
        store8<TO>(out + e, acc);
    }
}

// tf.image.resize nearest (v2, half-pixel): src = min(floor((dst+0.5)*in/out), in-1)   -- labels, int32
__global__ void resize_nearest_i32_kernel(const int32_t* __restrict__ x, int32_t* __restrict__ y, int N, int Hi, int Wi, int Ho,
                                          int Wo, int C, float sy, float sx) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const int iy = min((int)floorf(((float)oy + 0.5f) * sy), Hi - 1);
        const int ix = min((int)floorf(((float)ox + 0.5f) * sx), Wi - 1);
        y[i] = x[(((int64_t)n * Hi + iy) * Wi + ix) * C + c];
    }
}

static inline unsigned cap_blocks(int64_t items) {
    int64_t b = ceil_div64(items, 256);
    if (b > 256 * 8) b = 256 * 8;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int iseg_resize_bilinear_fwd(const void* x, int in_dtype, void* y, int out_dtype, int N, int Hi, int Wi, int Ho, int Wo,
                                        int C, hipStream_t stream) {
    ISEG_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "iseg_resize_bilinear_fwd: bad arguments");
    ISEG_REQUIRE((int64_t)Wo * C < (1ll << 30) && (int64_t)Wi * C < (1ll << 30) && (int64_t)N * Ho < (1ll << 31),
                 "iseg_resize_bilinear_fwd: row too long");
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    int64_t blocks = (int64_t)N * Ho;
    if (blocks > 256 * 16) blocks = 256 * 16;
#define RS(TI, TO)                                                                                                                    \
    hipLaunchKernelGGL((resize_bilinear_fwd_kernel<TI, TO>), dim3((unsigned)blocks), dim3(256), 0, stream, (const TI*)x, (TO*)y, N, Hi, \
                       Wi, Ho, Wo, C, sy, sx)
    const size_t lds = ((size_t)2 * Wi * C + (size_t)3 * Wo) * sizeof(float);
    if (lds <= 48 * 1024 && (int64_t)Wo * C >= 1024) {   // small source rows, long output rows: LDS-resident sources
#define RSL(TI, TO)                                                                                                                   \
    hipLaunchKernelGGL((resize_bilinear_fwd_lds_kernel<TI, TO>), dim3((unsigned)blocks), dim3(256), lds, stream, (const TI*)x, (TO*)y, \
                       N, Hi, Wi, Ho, Wo, C, sy, sx)
        if (in_dtype == ISEG_F32 && out_dtype == ISEG_F32) RSL(float, float);
        else if (in_dtype == ISEG_BF16 && out_dtype == ISEG_F32) RSL(bf16_t, float);
        else if (in_dtype == ISEG_BF16 && out_dtype == ISEG_BF16) RSL(bf16_t, bf16_t);
        else if (in_dtype == ISEG_F32 && out_dtype == ISEG_BF16) RSL(float, bf16_t);
        else {
            iseg_set_error("iseg_resize_bilinear_fwd: bad dtypes");
            return ISEG_ERR_ARG;
        }
#undef RSL
        return iseg_check_launch("iseg_resize_bilinear_fwd");
    }
    if (C % 8 == 0 && ((uintptr_t)x | (uintptr_t)y) % 16 == 0) {
        const unsigned vb = cap_blocks((int64_t)N * Ho * Wo * (C / 8));
#define RSV(TI, TO)                                                                                                                  \
    hipLaunchKernelGGL((resize_bilinear_fwd_vec_kernel<TI, TO>), dim3(vb), dim3(256), 0, stream, (const TI*)x, (TO*)y, N, Hi, Wi, Ho, \
                       Wo, C, sy, sx)
        if (in_dtype == ISEG_F32 && out_dtype == ISEG_F32) RSV(float, float);
        else if (in_dtype == ISEG_BF16 && out_dtype == ISEG_F32) RSV(bf16_t, float);
        else if (in_dtype == ISEG_BF16 && out_dtype == ISEG_BF16) RSV(bf16_t, bf16_t);
        else if (in_dtype == ISEG_F32 && out_dtype == ISEG_BF16) RSV(float, bf16_t);
        else {
            iseg_set_error("iseg_resize_bilinear_fwd: bad dtypes");
            return ISEG_ERR_ARG;
        }
#undef RSV
        return iseg_check_launch("iseg_resize_bilinear_fwd");
    }
    if (in_dtype == ISEG_F32 && out_dtype == ISEG_F32) RS(float, float);
    else if (in_dtype == ISEG_BF16 && out_dtype == ISEG_F32) RS(bf16_t, float);
    else if (in_dtype == ISEG_BF16 && out_dtype == ISEG_BF16) RS(bf16_t, bf16_t);
    else if (in_dtype == ISEG_F32 && out_dtype == ISEG_BF16) RS(float, bf16_t);
    else {
        iseg_set_error("iseg_resize_bilinear_fwd: bad dtypes");
        return ISEG_ERR_ARG;
    }
#undef RS
    return iseg_check_launch("iseg_resize_bilinear_fwd");
}

extern "C" int iseg_bn_relu_upsample_add(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const void* x,
                                         void* out, int N, int Hi, int Wi, int Ho, int Wo, int C, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(z && mean && rstd && gamma && beta && x && out && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0,
                 "iseg_bn_relu_upsample_add: bad arguments");
    ISEG_REQUIRE(C > 0 && C % 8 == 0, "iseg_bn_relu_upsample_add: C = %d must be a multiple of 8", C);
    ISEG_REQUIRE((int64_t)N * Ho * Wo < (1ll << 31), "iseg_bn_relu_upsample_add: more than 2^31 pixels");
    ISEG_REQUIRE((((uintptr_t)z | (uintptr_t)x | (uintptr_t)out | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
                 "iseg_bn_relu_upsample_add: operands must be 16-byte aligned");
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    // workgroup = tpc chunk lanes x rpi pixel lanes with no idle lane: tpc = the largest divisor-free fit of C / 8 into 256 (all of C / 8 when it
    // is at most 256), threads = tpc * (256 / tpc)
    const int C8 = C / 8;
    const int tpc = C8 < 256 ? C8 : 256;
    const int rpi = 256 / tpc;
    const int threads = tpc * rpi;
    int64_t blocks = ceil_div64((int64_t)N * Ho * Wo, (int64_t)rpi * 4);
    if (blocks > 4096) blocks = 4096;
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((bn_relu_upsample_add_kernel<bf16_t>), dim3((unsigned)blocks), dim3(threads), 0, stream, (const bf16_t*)z, mean, rstd, gamma,
                           beta, (const bf16_t*)x, (bf16_t*)out, N, Hi, Wi, Ho, Wo, C, sy, sx, tpc);
    else
        hipLaunchKernelGGL((bn_relu_upsample_add_kernel<float>), dim3((unsigned)blocks), dim3(threads), 0, stream, (const float*)z, mean, rstd, gamma,
                           beta, (const float*)x, (float*)out, N, Hi, Wi, Ho, Wo, C, sy, sx, tpc);
    return iseg_check_launch("iseg_bn_relu_upsample_add");
}

extern "C" size_t iseg_resize_bilinear_bwd_workspace_bytes(int N, int Hi, int Wi, int Ho, int Wo, int C) {
    (void)Hi;
    (void)Wo;
    return (size_t)N * Ho * Wi * C * sizeof(float);
}

// dy: [N,Ho,Wo,C] (dy_dtype) -> dx: [N,Hi,Wi,C] (dx_dtype); optional dx_add accumulates an existing gradient
extern "C" int iseg_resize_bilinear_bwd(const void* dy, int dy_dtype, void* dx, int dx_dtype, const void* dx_add, int N, int Hi,
                                        int Wi, int Ho, int Wo, int C, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(dy && dx, "iseg_resize_bilinear_bwd: null pointer");
    const size_t need = iseg_resize_bilinear_bwd_workspace_bytes(N, Hi, Wi, Ho, Wo, C);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_resize_bilinear_bwd: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    float* tmp = (float*)ws;
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    // exact x2 up-sampling with 16-byte channel chunks: one pass, no intermediate
    static const bool x2_off = [] { const char* e = getenv("ISEG_RESIZE_BWD_X2"); return e && atoi(e) == 0; }();
    if (!x2_off && Ho == 2 * Hi && Wo == 2 * Wi && C % 8 == 0 && dy_dtype == dx_dtype && (int64_t)N * Hi * Wi * (C / 8) < (1ll << 31) &&
        (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dx_add) & 15) == 0) {
        const unsigned vb = cap_blocks((int64_t)N * Hi * Wi * (C / 8));
        if (dy_dtype == ISEG_BF16)
            hipLaunchKernelGGL((resize_bwd_x2_vec_kernel<bf16_t, bf16_t>), dim3(vb), dim3(256), 0, stream, (const bf16_t*)dy, (bf16_t*)dx, N, Hi, Wi, C,
                               (const bf16_t*)dx_add);
        else
            hipLaunchKernelGGL((resize_bwd_x2_vec_kernel<float, float>), dim3(vb), dim3(256), 0, stream, (const float*)dy, (float*)dx, N, Hi, Wi, C,
                               (const float*)dx_add);
        return iseg_check_launch("iseg_resize_bilinear_bwd");
    }
    // X pass: [N*Ho, Wo, C] -> [N*Ho, Wi, C]
    const int64_t t1 = (int64_t)N * Ho * Wi * C;
    const size_t row_bytes = ((size_t)Wo * C + (size_t)3 * Wo) * sizeof(float);   // gradient row + destination lerp table
    if (row_bytes <= 64 * 1024 && (int64_t)N * Ho < (1ll << 31)) {
        int64_t blocks = (int64_t)N * Ho;
        if (blocks > 256 * 8) blocks = 256 * 8;
        if (dy_dtype == ISEG_BF16)
            hipLaunchKernelGGL((resize_bwd_x_lds_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), row_bytes, stream, (const bf16_t*)dy,
                               tmp, N * Ho, Wo, Wi, C, sx);
        else
            hipLaunchKernelGGL((resize_bwd_x_lds_kernel<float>), dim3((unsigned)blocks), dim3(256), row_bytes, stream, (const float*)dy,
                               tmp, N * Ho, Wo, Wi, C, sx);
    } else if (C % 8 == 0 && (uintptr_t)dy % 16 == 0 && dy_dtype == ISEG_BF16) {
        hipLaunchKernelGGL((resize_bwd_axis_vec_kernel<bf16_t, float>), dim3(cap_blocks(t1 / 8)), dim3(256), 0, stream, (const bf16_t*)dy,
                           tmp, (int64_t)N * Ho, Wo, Wi, (int64_t)C, sx, (const float*)nullptr);
    } else if (C % 8 == 0 && (uintptr_t)dy % 16 == 0) {
        hipLaunchKernelGGL((resize_bwd_axis_vec_kernel<float, float>), dim3(cap_blocks(t1 / 8)), dim3(256), 0, stream, (const float*)dy,
                           tmp, (int64_t)N * Ho, Wo, Wi, (int64_t)C, sx, (const float*)nullptr);
    } else if (dy_dtype == ISEG_BF16) {
        hipLaunchKernelGGL((resize_bwd_axis_kernel<bf16_t, float>), dim3(cap_blocks(t1)), dim3(256), 0, stream, (const bf16_t*)dy,
                           tmp, (int64_t)N * Ho, Wo, Wi, (int64_t)C, sx, (const float*)nullptr);
    } else {
        hipLaunchKernelGGL((resize_bwd_axis_kernel<float, float>), dim3(cap_blocks(t1)), dim3(256), 0, stream, (const float*)dy, tmp,
                           (int64_t)N * Ho, Wo, Wi, (int64_t)C, sx, (const float*)nullptr);
    }
    // Y pass: [N, Ho, Wi*C] -> [N, Hi, Wi*C]
    const int64_t t2 = (int64_t)N * Hi * Wi * C;
    const bool vec_y = ((int64_t)Wi * C) % 8 == 0 && ((uintptr_t)dx | (uintptr_t)dx_add) % 16 == 0;
    if (vec_y && dx_dtype == ISEG_BF16)
        hipLaunchKernelGGL((resize_bwd_axis_vec_kernel<float, bf16_t>), dim3(cap_blocks(t2 / 8)), dim3(256), 0, stream, (const float*)tmp,
                           (bf16_t*)dx, (int64_t)N, Ho, Hi, (int64_t)Wi * C, sy, (const bf16_t*)dx_add);
    else if (vec_y)
        hipLaunchKernelGGL((resize_bwd_axis_vec_kernel<float, float>), dim3(cap_blocks(t2 / 8)), dim3(256), 0, stream, (const float*)tmp,
                           (float*)dx, (int64_t)N, Ho, Hi, (int64_t)Wi * C, sy, (const float*)dx_add);
    else if (dx_dtype == ISEG_BF16)
        hipLaunchKernelGGL((resize_bwd_axis_kernel<float, bf16_t>), dim3(cap_blocks(t2)), dim3(256), 0, stream, (const float*)tmp,
                           (bf16_t*)dx, (int64_t)N, Ho, Hi, (int64_t)Wi * C, sy, (const bf16_t*)dx_add);
    else
        hipLaunchKernelGGL((resize_bwd_axis_kernel<float, float>), dim3(cap_blocks(t2)), dim3(256), 0, stream, (const float*)tmp,
                           (float*)dx, (int64_t)N, Ho, Hi, (int64_t)Wi * C, sy, (const float*)dx_add);
    return iseg_check_launch("iseg_resize_bilinear_bwd");
}

// tf.compat.v1.image.resize(x, size, method="bilinear", align_corners=True) (backbones/hrnet.py:303-304,523-524): generic gather kernels with
// the aligned-corner coordinate map; same lerp order as the half-pixel entry points
extern "C" int iseg_resize_bilinear_ac_fwd(const void* x, int in_dtype, void* y, int out_dtype, int N, int Hi, int Wi, int Ho, int Wo, int C,
                                           hipStream_t stream) {
    ISEG_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "iseg_resize_bilinear_ac_fwd: bad arguments");
    ISEG_REQUIRE((int64_t)Wo * C < (1ll << 30) && (int64_t)Wi * C < (1ll << 30) && (int64_t)N * Ho < (1ll << 31),
                 "iseg_resize_bilinear_ac_fwd: row too long");
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    int64_t blocks = (int64_t)N * Ho;
    if (blocks > 256 * 16) blocks = 256 * 16;
#define RSA(TI, TO)                                                                                                                         \
    hipLaunchKernelGGL((resize_bilinear_fwd_kernel<TI, TO, true>), dim3((unsigned)blocks), dim3(256), 0, stream, (const TI*)x, (TO*)y, N, Hi, \
                       Wi, Ho, Wo, C, sy, sx)
    if (in_dtype == ISEG_F32 && out_dtype == ISEG_F32) RSA(float, float);
    else if (in_dtype == ISEG_BF16 && out_dtype == ISEG_F32) RSA(bf16_t, float);
    else if (in_dtype == ISEG_BF16 && out_dtype == ISEG_BF16) RSA(bf16_t, bf16_t);
    else if (in_dtype == ISEG_F32 && out_dtype == ISEG_BF16) RSA(float, bf16_t);
    else {
        iseg_set_error("iseg_resize_bilinear_ac_fwd: bad dtypes");
        return ISEG_ERR_ARG;
    }
#undef RSA
    return iseg_check_launch("iseg_resize_bilinear_ac_fwd");
}

extern "C" int iseg_resize_bilinear_ac_bwd(const void* dy, int dy_dtype, void* dx, int dx_dtype, const void* dx_add, int N, int Hi, int Wi,
                                           int Ho, int Wo, int C, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(dy && dx && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "iseg_resize_bilinear_ac_bwd: bad arguments");
    const size_t need = iseg_resize_bilinear_bwd_workspace_bytes(N, Hi, Wi, Ho, Wo, C);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_resize_bilinear_ac_bwd: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    float* tmp = (float*)ws;
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const int64_t t1 = (int64_t)N * Ho * Wi * C, t2 = (int64_t)N * Hi * Wi * C;
    // X pass: [N*Ho, Wo, C] -> [N*Ho, Wi, C] (fp32), then Y pass: [N, Ho, Wi*C] -> [N, Hi, Wi*C]
    if (dy_dtype == ISEG_BF16)
        hipLaunchKernelGGL((resize_bwd_axis_kernel<bf16_t, float, true>), dim3(cap_blocks(t1)), dim3(256), 0, stream, (const bf16_t*)dy, tmp,
                           (int64_t)N * Ho, Wo, Wi, (int64_t)C, sx, (const float*)nullptr);
    else
        hipLaunchKernelGGL((resize_bwd_axis_kernel<float, float, true>), dim3(cap_blocks(t1)), dim3(256), 0, stream, (const float*)dy, tmp,
                           (int64_t)N * Ho, Wo, Wi, (int64_t)C, sx, (const float*)nullptr);
    if (dx_dtype == ISEG_BF16)
        hipLaunchKernelGGL((resize_bwd_axis_kernel<float, bf16_t, true>), dim3(cap_blocks(t2)), dim3(256), 0, stream, (const float*)tmp,
                           (bf16_t*)dx, (int64_t)N, Ho, Hi, (int64_t)Wi * C, sy, (const bf16_t*)dx_add);
    else
        hipLaunchKernelGGL((resize_bwd_axis_kernel<float, float, true>), dim3(cap_blocks(t2)), dim3(256), 0, stream, (const float*)tmp,
                           (float*)dx, (int64_t)N, Ho, Hi, (int64_t)Wi * C, sy, (const float*)dx_add);
    return iseg_check_launch("iseg_resize_bilinear_ac_bwd");
}

extern "C" int iseg_resize_nearest_i32(const int32_t* x, int32_t* y, int N, int Hi, int Wi, int Ho, int Wo, int C,
                                       hipStream_t stream) {
    ISEG_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "iseg_resize_nearest_i32: bad arguments");
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    hipLaunchKernelGGL(resize_nearest_i32_kernel, dim3(cap_blocks((int64_t)N * Ho * Wo * C)), dim3(256), 0, stream, x, y, N, Hi, Wi,
                       Ho, Wo, C, sy, sx);
    return iseg_check_launch("iseg_resize_nearest_i32");
}
