// Depthwise KxK convolution on NHWC tensors (keras.layers.DepthwiseConv2D, padding="same", stride 1,
// dilation d) -- the 7x7 depthwise of backbones/convnext.py:25,50 (dilated by build_dilated_convnext :245-266)
// and the 3x3 depthwise of layers/dcn_v3/dcn_v3.py.  49 FMA per element but only 4 B of HBM traffic: the kernels
// are limited by how many L1/L2 loads they keep in flight, so:
//   * a lane owns CV (4 or 8) consecutive channels (one 8/16-B load per pixel) and TW output pixels along W, and slides a
//     register window over the input row so each loaded pixel feeds up to KW taps;
//   * every load is issued unconditionally from a clamped address and zeroed by a select afterwards (no divergent
//     branch between loads, so a whole kernel row's loads are in flight together); interior tiles skip the selects;
//   * weights of the block's channel slab sit in LDS as fp32 (the fp32 master kernel is read directly);
//   * backward-data is the same kernel with the taps flipped and the complementary padding, and can add the
//     residual branch's gradient on the way out (dx = dres + dwconv^T(dy));
//   * backward-weight gives each lane one kernel row (KW taps x CV channels of accumulators) and a run of
//     (image row, W-segment) items, slides a KW-wide register window along W, then reduces lanes -> block (LDS) ->
//     grid (fixed-order partial sums, deterministic).
#include "common.h"
#include "iseg_hip.h"
#include <stdlib.h>

namespace {

constexpr int TW = 4;
constexpr int BWSEG = 32;  // widest W segment the register-batched weight-gradient path holds

template <class T, int CV> __device__ __forceinline__ void loadv(const T* p, float* out);
template <> __device__ __forceinline__ void loadv<float, 8>(const float* p, float* out) { load8<float>(p, out); }
template <> __device__ __forceinline__ void loadv<bf16_t, 8>(const bf16_t* p, float* out) { load8<bf16_t>(p, out); }
template <> __device__ __forceinline__ void loadv<float, 4>(const float* p, float* out) { Vec16<float>::load(p, out); }
template <> __device__ __forceinline__ void loadv<bf16_t, 4>(const bf16_t* p, float* out) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = (float)v[i];
}
template <class T, int CV> __device__ __forceinline__ void storev(T* p, const float* in);
template <> __device__ __forceinline__ void storev<float, 8>(float* p, const float* in) { store8<float>(p, in); }
template <> __device__ __forceinline__ void storev<bf16_t, 8>(bf16_t* p, const float* in) { store8<bf16_t>(p, in); }
template <> __device__ __forceinline__ void storev<float, 4>(float* p, const float* in) { Vec16<float>::store(p, in); }
template <> __device__ __forceinline__ void storev<bf16_t, 4>(bf16_t* p, const float* in) {
    bf16x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (bf16_t)in[i];
    *reinterpret_cast<bf16x4*>(p) = v;
}

// channel groups (of CV channels) per block slab: the largest divisor of C/CV that is <= maxg
static inline int groups_per_slab(int C, int CV, int maxg) {
    const int G = C / CV;
    int best = 1;
    for (int g = 1; g <= maxg && g <= G; ++g)
        if (G % g == 0) best = g;
    return best;
}

// unconditional load from a clamped pixel index, zeroed when the true index is outside [0, W)
template <class T, int CV, bool CHECK>
__device__ __forceinline__ void load_px(const T* __restrict__ row, int iw, int W, int C, bool row_ok, float* out) {
    if (CHECK) {
        const bool ok = row_ok && (unsigned)iw < (unsigned)W;
        const int iwc = min(max(iw, 0), W - 1);
        loadv<T, CV>(row + iwc * C, out);
#pragma unroll
        for (int u = 0; u < CV; ++u) out[u] = ok ? out[u] : 0.f;
    } else {
        loadv<T, CV>(row + iw * C, out);
    }
}

template <class T, int K, int CV, bool DIL1, bool CHECK>
__device__ __forceinline__ void dw_accumulate_row(const T* __restrict__ xr, const float* __restrict__ wrow, int w0, int pad_l, int W,
                                                  int C, int dil, int sc, float (&acc)[TW][CV]) {
    if (DIL1) {
        float xin[TW + K - 1][CV];
#pragma unroll
        for (int s = 0; s < TW + K - 1; ++s) load_px<T, CV, CHECK>(xr, w0 - pad_l + s, W, C, true, xin[s]);
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            float wv[CV];
            loadv<float, CV>(wrow + kw * sc, wv);
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int u = 0; u < CV; ++u) acc[t][u] = fmaf(xin[t + kw][u], wv[u], acc[t][u]);
        }
    } else {
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            float wv[CV];
            loadv<float, CV>(wrow + kw * sc, wv);
            float xv[TW][CV];
#pragma unroll
            for (int t = 0; t < TW; ++t) load_px<T, CV, true>(xr, w0 + t + kw * dil - pad_l, W, C, true, xv[t]);
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int u = 0; u < CV; ++u) acc[t][u] = fmaf(xv[t][u], wv[u], acc[t][u]);
        }
    }
}

// ROLLED: keep the kernel-row loop rolled so only one row's loads are live (fewer VGPRs, more waves per SIMD)
template <class T, int K, int CV, bool DIL1, bool ROLLED>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const T* __restrict__ add,
                                                         T* __restrict__ y, int N, int H, int W, int C, int dil, int pad_t,
                                                         int pad_l, int flip, int gs, int pt) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [K*K][gs*CV]
    const int sc = gs * CV;
    const int slab_c0 = blockIdx.y * sc;
    for (int i = threadIdx.x; i < K * K * sc; i += blockDim.x) {
        int tap = i / sc;
        const int c = i % sc;
        if (flip) tap = K * K - 1 - tap;
        wl[i] = w[tap * C + slab_c0 + c];
    }
    __syncthreads();
    const int cg = threadIdx.x % gs, ptile = threadIdx.x / gs;
    if (ptile >= pt) return;
    const int wtiles = (W + TW - 1) / TW;
    const int tiles_total = N * H * wtiles;
    const int c0 = slab_c0 + cg * CV;
    float bv[CV];
#pragma unroll
    for (int u = 0; u < CV; ++u) bv[u] = bias ? bias[c0 + u] : 0.f;

    for (int tile = blockIdx.x * pt + ptile; tile < tiles_total; tile += gridDim.x * pt) {
        const int wt = tile % wtiles;
        const int nh = tile / wtiles;
        const int h = nh % H;
        const int n = nh / H;
        const int w0 = wt * TW;
        float acc[TW][CV];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int u = 0; u < CV; ++u) acc[t][u] = bv[u];
        const T* xn = x + (int64_t)n * H * W * C + c0;
        const bool interior = DIL1 && (w0 - pad_l >= 0) && (w0 - pad_l + TW + K - 2 < W);
        auto row = [&](int kh) {
            const int ih = h + kh * dil - pad_t;
            if ((unsigned)ih < (unsigned)H) {
                const T* xr = xn + ih * W * C;
                const float* wrow = wl + kh * K * sc + cg * CV;
                if (interior) dw_accumulate_row<T, K, CV, DIL1, false>(xr, wrow, w0, pad_l, W, C, dil, sc, acc);
                else dw_accumulate_row<T, K, CV, DIL1, true>(xr, wrow, w0, pad_l, W, C, dil, sc, acc);
            }
        };
        if (ROLLED) {
#pragma unroll 1
            for (int kh = 0; kh < K; ++kh) row(kh);
        } else {
#pragma unroll
            for (int kh = 0; kh < K; ++kh) row(kh);
        }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int ow = w0 + t;
            if (ow < W) {
                const int64_t off = (((int64_t)n * H + h) * W + ow) * C + c0;
                if (add) {
                    float a[CV];
                    loadv<T, CV>(add + off, a);
#pragma unroll
                    for (int u = 0; u < CV; ++u) acc[t][u] += a[u];
                }
                storev<T, CV>(y + off, acc[t]);
            }
        }
    }
}

// LDS-tiled forward / data-gradient: a workgroup stages the (TH + halo) x (16 + halo) input patch of its channel slab once
// (coalesced 16-B loads, zero-filled outside the image), then every lane computes 4 pixels x 8 channels from LDS.
// Replaces ~70 L1/L2 loads per lane by one cooperative patch load (halo read amplification ~2.4x, served by L2).
constexpr int TWB = 16;  // output columns per workgroup tile (4 lanes x TW)

template <class T, int K, bool DIL1>
__global__ __launch_bounds__(256) void dwconv_fwd_lds_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const T* __restrict__ add,
                                                             T* __restrict__ y, int N, int H, int W, int C, int dil, int pad_t,
                                                             int pad_l, int flip, int gs, int TH, int tiles_h, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) char smem_dw[];
    const int sc = gs * 8;
    const int halo = (K - 1) * dil;
    const int IH = TH + halo, IW = TWB + halo;
    float* wl = reinterpret_cast<float*>(smem_dw);                        // [K*K][sc] fp32
    T* xt = reinterpret_cast<T*>(smem_dw + (size_t)K * K * sc * sizeof(float));  // [IH][IW][scp]
    const int scp = sc + (int)(16 / sizeof(T));  // pixel stride padded by one 16-B slot: neighbouring 4-pixel tiles hit other banks
    const int slab_c0 = blockIdx.y * sc;
    int b = blockIdx.x;
    const int tw_i = b % tiles_w;
    b /= tiles_w;
    const int th_i = b % tiles_h;
    const int n = b / tiles_h;
    const int h0 = th_i * TH, w0 = tw_i * TWB;
    // Both fills are written as "issue a batch of independent loads, then store the batch": with a plain strided loop hipcc
    // waits for every load before issuing the next one, and the ~12 dependent L2 round trips per lane dominated the kernel.
    {
        const int nw4 = K * K * sc / 4;  // weights as float4 (sc is a multiple of 8)
        constexpr int WB = 5;
        for (int base = threadIdx.x; base < nw4; base += 256 * WB) {
            float4 wr[WB];
#pragma unroll
            for (int q = 0; q < WB; ++q) {
                const int i = base + q * 256;
                if (i < nw4) {
                    int tap = (i * 4) / sc;
                    const int c = (i * 4) % sc;
                    if (flip) tap = K * K - 1 - tap;
                    wr[q] = *reinterpret_cast<const float4*>(w + tap * C + slab_c0 + c);
                }
            }
#pragma unroll
            for (int q = 0; q < WB; ++q) {
                const int i = base + q * 256;
                if (i < nw4) reinterpret_cast<float4*>(wl)[i] = wr[q];
            }
        }
    }
    const T* xn = x + (int64_t)n * H * W * C + slab_c0;
    const int nchunks = IH * IW * gs;
    constexpr int XB = (sizeof(T) == 2) ? 12 : 6;  // 16-B (bf16) / 32-B (fp32) chunks in flight per lane
    for (int base = threadIdx.x; base < nchunks; base += 256 * XB) {
        float4 ra[XB], rb[XB];
        int pcs[XB];
#pragma unroll
        for (int q = 0; q < XB; ++q) {
            const int i = base + q * 256;
            pcs[q] = -1;
            if (i < nchunks) {
                const int g = i % gs;
                const int pc = i / gs;
                const int c = pc % IW, r = pc / IW;
                const int ih = h0 - pad_t + r, iw = w0 - pad_l + c;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                const T* src = xn + ((ok ? ih : 0) * W + (ok ? iw : 0)) * C + g * 8;
                ra[q] = *reinterpret_cast<const float4*>(src);
                if (sizeof(T) == 4) rb[q] = *reinterpret_cast<const float4*>(src + 4);
                if (!ok) ra[q] = rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                pcs[q] = pc * scp + g * 8;
            }
        }
#pragma unroll
        for (int q = 0; q < XB; ++q) {
            if (pcs[q] >= 0) {
                T* dst = xt + pcs[q];
                *reinterpret_cast<float4*>(dst) = ra[q];
                if (sizeof(T) == 4) *reinterpret_cast<float4*>(dst + 4) = rb[q];
            }
        }
    }
    __syncthreads();
    const int cg = threadIdx.x % gs, pt = threadIdx.x / gs;
    const int ty = pt / (TWB / TW), tx = pt % (TWB / TW);
    if (ty >= TH) return;
    const int oh = h0 + ty;
    if (oh >= H) return;
    const int c0 = slab_c0 + cg * 8;
    float acc[TW][8];
    {
        float bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bv[u] = bias ? bias[c0 + u] : 0.f;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[t][u] = bv[u];
    }
#pragma unroll 1  // rolled on purpose: unrolled, hipcc hoists all 70 LDS reads and spills 400 VGPRs to scratch
    for (int kh = 0; kh < K; ++kh) {
        const T* xr = xt + (size_t)((ty + kh * dil) * IW + tx * TW) * scp + cg * 8;
        const float* wrow = wl + kh * K * sc + cg * 8;
        if (DIL1) {
            float xin[TW + K - 1][8];
#pragma unroll
            for (int s = 0; s < TW + K - 1; ++s) load8<T>(xr + s * scp, xin[s]);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                float wv[8];
                load8<float>(wrow + kw * sc, wv);
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[t][u] = fmaf(xin[t + kw][u], wv[u], acc[t][u]);
            }
        } else {
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                float wv[8];
                load8<float>(wrow + kw * sc, wv);
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    float xv[8];
                    load8<T>(xr + (t + kw * dil) * scp, xv);
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[t][u] = fmaf(xv[u], wv[u], acc[t][u]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int ow = w0 + tx * TW + t;
        if (ow < W) {
            const int64_t off = (((int64_t)n * H + oh) * W + ow) * C + c0;
            if (add) {
                float a[8];
                load8<T>(add + off, a);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[t][u] += a[u];
            }
            store8<T>(y + off, acc[t]);
        }
    }
}

// dw[kh][kw][c] = sum_{n,h,w} x[n, h+kh*d-pt, w+kw*d-pl, c] * dy[n,h,w,c];  db[c] = sum dy
// thread = (channel group, kernel row kh, item lane); an item is (image row, W segment of `wseg` pixels).
template <class T, int K, int CV, bool DIL1>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                float* __restrict__ partials, int N, int H, int W, int C,
                                                                int dil, int pad_t, int pad_l, int gs, int rt, int wseg, int ipl) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [rt][(K*K+1)][gs*CV]: one slab per item lane, summed in lane order
    const int sc = gs * CV;
    const int slab_c0 = blockIdx.y * sc;
    const int nred = (K * K + 1) * sc;
    const int cg = threadIdx.x % gs;
    const int kh = (threadIdx.x / gs) % K;
    const int rl = threadIdx.x / (gs * K);
    if (rl < rt) {
        const int c0 = slab_c0 + cg * CV;
        float acc[K][CV], accb[CV];
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int u = 0; u < CV; ++u) acc[j][u] = 0.f;
#pragma unroll
        for (int u = 0; u < CV; ++u) accb[u] = 0.f;
        const int nseg = (W + wseg - 1) / wseg;
        const int items = N * H * nseg;
        for (int q = 0; q < ipl; ++q) {
            const int item = (blockIdx.x * ipl + q) * rt + rl;
            if (item >= items) break;
            const int seg = item % nseg;
            const int nh = item / nseg;
            const int h = nh % H, n = nh / H;
            const int ws = seg * wseg;
            const int we = min(W, ws + wseg);
            const int ih = h + kh * dil - pad_t;
            const bool row_ok = (unsigned)ih < (unsigned)H;
            const T* dyr = dy + (((int64_t)n * H + h) * W) * C + c0;
            const T* xr = x + (((int64_t)n * H + (row_ok ? ih : 0)) * W) * C + c0;
            if (DIL1 && sizeof(T) == 2 && CV == 4 && wseg <= BWSEG) {
                // Small planes (W <= 32, stages 3-4 of ConvNeXt): the whole row segment is requested up front as packed 8-byte
                // pixels (<= 38 + 32 loads in flight per lane) and the window slides over registers.  With one x and one dy load per
                // pixel inside the sliding loop every step waited a full L2 round trip: 81 us per launch at 16x32x32x384.
                uint2 xraw[BWSEG + K - 1], draw[BWSEG];
                const int cnt = we - ws;
#pragma unroll
                for (int t = 0; t < BWSEG + K - 1; ++t) {
                    const int iw = ws - pad_l + t;
                    const bool ok = row_ok && t < cnt + K - 1 && (unsigned)iw < (unsigned)W;
                    const uint2 v = *reinterpret_cast<const uint2*>(xr + (int64_t)min(max(iw, 0), W - 1) * C);
                    xraw[t] = ok ? v : make_uint2(0u, 0u);
                }
#pragma unroll
                for (int t = 0; t < BWSEG; ++t) {
                    const bool ok = t < cnt;
                    const uint2 v = *reinterpret_cast<const uint2*>(dyr + (int64_t)(ok ? ws + t : ws) * C);
                    draw[t] = ok ? v : make_uint2(0u, 0u);
                }
                auto unpack = [](const uint2& r, float* o) {
                    o[0] = __uint_as_float(r.x << 16);
                    o[1] = __uint_as_float(r.x & 0xffff0000u);
                    o[2] = __uint_as_float(r.y << 16);
                    o[3] = __uint_as_float(r.y & 0xffff0000u);
                };
                float win[K][CV];
#pragma unroll
                for (int j = 0; j < K - 1; ++j) unpack(xraw[j], win[j]);
#pragma unroll
                for (int t = 0; t < BWSEG; ++t) {
                    unpack(xraw[t + K - 1], win[(t + K - 1) % K]);
                    float d[CV];
                    unpack(draw[t], d);
                    if (kh == 0) {
#pragma unroll
                        for (int u = 0; u < CV; ++u) accb[u] += d[u];
                    }
#pragma unroll
                    for (int j = 0; j < K; ++j)
#pragma unroll
                        for (int u = 0; u < CV; ++u) acc[j][u] = fmaf(win[(t + j) % K][u], d[u], acc[j][u]);
                }
            } else if (DIL1) {
                float win[K][CV];
#pragma unroll
                for (int j = 0; j < K - 1; ++j) load_px<T, CV, true>(xr, ws - pad_l + j, W, C, row_ok, win[j]);
                for (int base = ws; base < we; base += K) {
#pragma unroll
                    for (int t = 0; t < K; ++t) {
                        const int ow = base + t;
                        load_px<T, CV, true>(xr, ow - pad_l + (K - 1), W, C, row_ok, win[(t + K - 1) % K]);
                        float d[CV];
                        load_px<T, CV, true>(dyr, ow < we ? ow : -1, W, C, true, d);
                        if (kh == 0) {
#pragma unroll
                            for (int u = 0; u < CV; ++u) accb[u] += d[u];
                        }
#pragma unroll
                        for (int j = 0; j < K; ++j)
#pragma unroll
                            for (int u = 0; u < CV; ++u) acc[j][u] = fmaf(win[(t + j) % K][u], d[u], acc[j][u]);
                    }
                }
            } else {
                for (int ow = ws; ow < we; ++ow) {
                    float d[CV];
                    loadv<T, CV>(dyr + ow * C, d);
                    if (kh == 0) {
#pragma unroll
                        for (int u = 0; u < CV; ++u) accb[u] += d[u];
                    }
                    float xv[K][CV];
#pragma unroll
                    for (int j = 0; j < K; ++j) load_px<T, CV, true>(xr, ow + j * dil - pad_l, W, C, row_ok, xv[j]);
#pragma unroll
                    for (int j = 0; j < K; ++j)
#pragma unroll
                        for (int u = 0; u < CV; ++u) acc[j][u] = fmaf(xv[j][u], d[u], acc[j][u]);
                }
            }
        }
        // no float atomics (the step is bit-reproducible, core_env.py:39-48): lane (cg, kh, rl) owns K taps of slab rl -- every cell of the
        // rt slabs is written exactly once (lanes that ran out of items store their zeros) -- and the slabs are summed in lane order below
        float* mine = red + (size_t)rl * nred;
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int u = 0; u < CV; ++u) mine[(kh * K + j) * sc + cg * CV + u] = acc[j][u];
        if (kh == 0) {
#pragma unroll
            for (int u = 0; u < CV; ++u) mine[K * K * sc + cg * CV + u] = accb[u];
        }
    }
    __syncthreads();
    // partial layout: [block.x][(K*K+1)][C]
    float* out = partials + (int64_t)blockIdx.x * (K * K + 1) * C;
    for (int i = threadIdx.x; i < nred; i += blockDim.x) {
        const int tap = i / sc, c = i % sc;
        float a = red[i];
        for (int r = 1; r < rt; ++r) a += red[(size_t)r * nred + i];
        out[(int64_t)tap * C + slab_c0 + c] = a;
    }
}

// LDS-tiled weight gradient (dil == 1): persistent workgroups walk (image, row-band, column-band) tiles of a channel slab;
// the dy tile and the x tile (with halo) are staged once per tile, lane (cg, kh, rl) slides a K-wide window along row rl of
// the tile for kernel row kh, accumulators live in registers across all tiles of the workgroup.
constexpr int BWW = 32;  // tile columns

template <class T, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_lds_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                    float* __restrict__ partials, int N, int H, int W, int C,
                                                                    int pad_t, int pad_l, int gs, int rt, int tiles_h, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) char smem_bw[];
    const int sc = gs * 8;
    const int TH = rt;
    const int IH = TH + K - 1, IW = BWW + K - 1;
    T* xt = reinterpret_cast<T*>(smem_bw);                                   // [IH][IW][sc]
    T* dt = xt + (size_t)IH * IW * sc;                                       // [TH][BWW][sc]
    float* red = reinterpret_cast<float*>(smem_bw);                          // [rt][(K*K+1)][sc]: over the tiles once the last one is consumed
    const int slab_c0 = blockIdx.y * sc;
    const int nred = (K * K + 1) * sc;
    const int cg = threadIdx.x % gs;
    const int kh = (threadIdx.x / gs) % K;
    const int rl = threadIdx.x / (gs * K);
    const bool worker = rl < rt;
    float acc[K][8], accb[8];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[j][u] = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) accb[u] = 0.f;
    const int ntiles = N * tiles_h * tiles_w;
    constexpr int XB = (sizeof(T) == 2) ? 8 : 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tw_i = b % tiles_w;
        b /= tiles_w;
        const int th_i = b % tiles_h;
        const int n = b / tiles_h;
        const int h0 = th_i * TH, w0 = tw_i * BWW;
        __syncthreads();  // previous tile consumed
        const T* xn = x + (int64_t)n * H * W * C + slab_c0;
        const T* dn = dy + (int64_t)n * H * W * C + slab_c0;
        const int nx = IH * IW * gs, nd = TH * BWW * gs;
        for (int base = threadIdx.x; base < nx + nd; base += 256 * XB) {
            float4 ra[XB], rb[XB];
            int dsts[XB];
#pragma unroll
            for (int q = 0; q < XB; ++q) {
                int i = base + q * 256;
                dsts[q] = -1;
                if (i < nx + nd) {
                    const bool is_x = i < nx;
                    if (!is_x) i -= nx;
                    const int g = i % gs;
                    const int pc = i / gs;
                    const int cw = is_x ? IW : BWW;
                    const int c = pc % cw, r = pc / cw;
                    const int ih = is_x ? h0 - pad_t + r : h0 + r;
                    const int iw = is_x ? w0 - pad_l + c : w0 + c;
                    const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    const T* src = (is_x ? xn : dn) + ((ok ? ih : 0) * W + (ok ? iw : 0)) * C + g * 8;
                    ra[q] = *reinterpret_cast<const float4*>(src);
                    if (sizeof(T) == 4) rb[q] = *reinterpret_cast<const float4*>(src + 4);
                    if (!ok) ra[q] = rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    dsts[q] = (is_x ? 0 : IH * IW * sc) + pc * sc + g * 8;
                }
            }
#pragma unroll
            for (int q = 0; q < XB; ++q) {
                if (dsts[q] >= 0) {
                    T* dst = xt + dsts[q];
                    *reinterpret_cast<float4*>(dst) = ra[q];
                    if (sizeof(T) == 4) *reinterpret_cast<float4*>(dst + 4) = rb[q];
                }
            }
        }
        __syncthreads();
        if (worker) {
            const T* xr = xt + (size_t)((rl + kh) * IW) * sc + cg * 8;   // x row (rl + kh) of the halo tile
            const T* dr = dt + (size_t)(rl * BWW) * sc + cg * 8;
            float win[K][8];
#pragma unroll
            for (int j = 0; j < K - 1; ++j) load8<T>(xr + j * sc, win[j]);
#pragma unroll 1
            for (int base = 0; base < BWW; base += K) {
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    const int ow = base + t;
                    if (ow < BWW) {   // uniform: BWW and K are compile-time
                        load8<T>(xr + (ow + K - 1) * sc, win[(t + K - 1) % K]);
                        float d[8];
                        load8<T>(dr + ow * sc, d);
                        if (kh == 0) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) accb[u] += d[u];
                        }
#pragma unroll
                        for (int j = 0; j < K; ++j)
#pragma unroll
                            for (int u = 0; u < 8; ++u) acc[j][u] = fmaf(win[(t + j) % K][u], d[u], acc[j][u]);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (worker) {      // one slab per tile-row lane, every cell written once, summed in lane order (no float atomics: bit-reproducible)
        float* mine = red + (size_t)rl * nred;
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int u = 0; u < 8; ++u) mine[(kh * K + j) * sc + cg * 8 + u] = acc[j][u];
        if (kh == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) mine[K * K * sc + cg * 8 + u] = accb[u];
        }
    }
    __syncthreads();
    float* out = partials + (int64_t)blockIdx.x * (K * K + 1) * C;
    for (int i = threadIdx.x; i < nred; i += 256) {
        const int tap = i / sc, c = i % sc;
        float a = red[i];
        for (int r = 1; r < rt; ++r) a += red[(size_t)r * nred + i];
        out[(int64_t)tap * C + slab_c0 + c] = a;
    }
}

// experiment knobs (read once): ISEG_DW_FWD_CV / ISEG_DW_BW_CV in {4,8}, ISEG_DW_FWD_ROLLED in {0,1}
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static int fwd_cv() { static int v = env_int("ISEG_DW_FWD_CV", 8); return v == 4 ? 4 : 8; }
static int fwd_rolled() { static int v = env_int("ISEG_DW_FWD_ROLLED", 1); return v != 0; }
static int bw_cv() { static int v = env_int("ISEG_DW_BW_CV", 4); return v == 8 ? 8 : 4; }

// DMA-tiled bf16 weight gradient (K x K, dil == 1, C % 32 == 0): the variant the ConvNeXt stages take.
//   * a workgroup owns a 32-channel slab and walks (image, row band, column band) tiles; the x tile (with halo) and the dy tile
//     go HBM -> LDS by global_load_lds_dwordx4 (one instruction = 16 pixels x 64 B; halo and padding pixels read a zero page), so
//     the fill costs one address per 16-B piece and no staging registers; two workgroups per CU overlap one's fill with the
//     other's arithmetic;
//   * lane (cg, r, ky): 8 channels, tile row r, kernel row ky -- slides a K-wide register window along the row: per output pixel
//     two ds_read_b128, 16 unpack and 4 K packed FMAs (K x 8 MACs); ky is the slowest lane index so only the first wavefront
//     carries the bias sums;
//   * row strides of 39 / 33 pixels (== 192 / 64 mod 256 B) keep the four rows a 16-lane read group touches on distinct banks;
//   * accumulators persist over the workgroup's tiles; lanes -> workgroup by a fixed-order sum over r in LDS, workgroups -> result
//     by launch_reduce_rows (deterministic).
// Measured (16 images, us per launch incl. the partial reduce, register-batched kernel -> this one): 128x128x96 133.5 -> 70.4, 64x64x192
// 71.3 -> 42.8, 32x32x384 44.8 -> 25.2, 16x16x768 42.7 -> 17.5.  SQ counters at 128x128x96: 20.1 M VALU wave-instructions (1740 per tile
// and wavefront: 32 steps x (28 packed FMA + 16 unpack) + ~285 for the fill), VALU busy 78 % of the wavefront lifetime -- the kernel is
// bound by the unpack + FMA issue, not by LDS (bank conflicts 5 % of LDS cycles) or HBM (L2 hit rate 80 %, 14 MB of misses).
typedef __attribute__((address_space(3))) void* dw_lds_ptr;
typedef const __attribute__((address_space(1))) void* dw_glb_ptr;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ uint4 dw_zero_page[4];      // 64 zero bytes (device globals are zero-initialised)

__device__ __forceinline__ void unpack_bf16x8(const char* p, f32x2* out) {
    const bf16x8 raw = *reinterpret_cast<const bf16x8*>(p);
    const uint4 v = __builtin_bit_cast(uint4, raw);
    out[0] = f32x2{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
    out[1] = f32x2{__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
    out[2] = f32x2{__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u)};
    out[3] = f32x2{__uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)};
}

template <int K, int TWD>
__global__ __launch_bounds__(256, 2) void dwconv_bwd_weight_dma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                       float* __restrict__ partials, int N, int H, int W, int C,
                                                                       int pad_t, int pad_l, int TH, int tiles_h, int tiles_w) {
    constexpr int IW = TWD + K - 1, IWP = IW + 1, DWP = TWD + 1, NT = K * K + 1;
    static_assert((IWP * 64) % 256 == 192 && (DWP * 64) % 256 == 64, "row strides must spread rows over the banks");
    extern __shared__ __attribute__((aligned(1024))) char smem_wg[];
    const int IH = TH + K - 1;
    const int xpieces = (IH * IWP + 15) / 16, dpieces = (TH * DWP + 15) / 16;
    char* xt = smem_wg;                      // [IH][IWP][32] bf16
    char* dt = smem_wg + xpieces * 1024;     // [TH][DWP][32] bf16
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = tid & 3, r = (tid >> 2) % TH, ky = (tid >> 2) / TH;
    const bool worker = ky < K;
    const int c0 = blockIdx.y * 32;
    // blockIdx.x round-robins over the 8 XCDs: give each XCD a contiguous run of the tile sequence (neighbouring tiles share halo in its L2)
    int lb = blockIdx.x;
    if (gridDim.x % 8 == 0) lb = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    f32x2 acc[K][4], accb[4];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = f32x2{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) accb[q] = f32x2{0.f, 0.f};
    const int ntiles = N * tiles_h * tiles_w;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(dw_zero_page) + (lane & 3) * 8;
    // DMA piece i of this wavefront is x piece (dy piece) wid + 4 i of the tile image; its pixel's (row, column) within the tile does not
    // depend on the tile, so it is worked out once.  Padding slots (the spare column, rows past the tile) get row 0x7fff: never inside
    // the image.
    constexpr int MAXPX = (((9 + K - 1) * IWP + 15) / 16 + 3) / 4, MAXPD = ((9 * DWP + 15) / 16 + 3) / 4;
    int xrc[MAXPX], drc[MAXPD];
#pragma unroll
    for (int i = 0; i < MAXPX; ++i) {
        const int pix = (wid + 4 * i) * 16 + (lane >> 2);
        const int rr = pix / IWP, cc = pix - rr * IWP;
        xrc[i] = ((cc >= IW || rr >= IH ? 0x7fff : rr) << 16) | cc;
    }
#pragma unroll
    for (int i = 0; i < MAXPD; ++i) {
        const int pix = (wid + 4 * i) * 16 + (lane >> 2);
        const int rr = pix / DWP, cc = pix - rr * DWP;
        drc[i] = ((cc >= TWD || rr >= TH ? 0x7fff : rr) << 16) | cc;
    }
    for (int tile = lb; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tw_i = b % tiles_w;
        b /= tiles_w;
        const int th_i = b % tiles_h;
        const int n = b / tiles_h;
        const int h0 = th_i * TH, w0 = tw_i * TWD;
        __syncthreads();      // previous tile consumed
        const bf16_t* xb = x + (((n * H + h0 - pad_t) * W + w0 - pad_l) * C + c0 + (lane & 3) * 8);      // (may point before the image: only used under `ok`)
        const bf16_t* db = dy + (((n * H + h0) * W + w0) * C + c0 + (lane & 3) * 8);
#pragma unroll
        for (int i = 0; i < MAXPX; ++i) {
            const int p = wid + 4 * i;
            if (p < xpieces) {      // wavefront-uniform
                int rc = xrc[i];
                asm volatile("" : "+v"(rc));      // keep one register per piece: without this hipcc hoists every derived offset out of the tile loop and spills
                const int rr = rc >> 16, cc = rc & 0xffff;
                const bool ok = (unsigned)(h0 - pad_t + rr) < (unsigned)H && (unsigned)(w0 - pad_l + cc) < (unsigned)W;
                const bf16_t* src = ok ? xb + __mul24(__mul24(rr, W) + cc, C) : zero;
                __builtin_amdgcn_global_load_lds((dw_glb_ptr)src, (dw_lds_ptr)(xt + p * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < MAXPD; ++i) {
            const int p = wid + 4 * i;
            if (p < dpieces) {
                int rc = drc[i];
                asm volatile("" : "+v"(rc));
                const int rr = rc >> 16, cc = rc & 0xffff;
                const bool ok = (unsigned)(h0 + rr) < (unsigned)H && (unsigned)(w0 + cc) < (unsigned)W;
                const bf16_t* src = ok ? db + __mul24(__mul24(rr, W) + cc, C) : zero;
                __builtin_amdgcn_global_load_lds((dw_glb_ptr)src, (dw_lds_ptr)(dt + p * 1024), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (worker) {
            const char* xr = xt + ((r + ky) * IWP) * 64 + cg * 16;
            const char* dr = dt + (r * DWP) * 64 + cg * 16;
            f32x2 win[K][4];
#pragma unroll
            for (int j = 0; j < K - 1; ++j) unpack_bf16x8(xr + j * 64, win[j]);
            // one output pixel: window slot (t + K - 1) % K takes the new x pixel, tap j reads slot (t + j) % K.  t only enters through
            // t % K, so the row runs as a rolled loop over groups of K pixels plus a static tail (a full unroll makes hipcc hoist
            // every ds_read of the row and spill)
            auto step = [&](const char* xp, const char* dp, int t) {
                unpack_bf16x8(xp, win[(t + K - 1) % K]);
                f32x2 d[4];
                unpack_bf16x8(dp, d);
                if (wid == 0) {      // kernel row 0 lives in the first wavefront only (ky is the slowest lane index)
#pragma unroll
                    for (int q = 0; q < 4; ++q) accb[q] += d[q];
                }
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[j][q] = __builtin_elementwise_fma(win[(t + j) % K][q], d[q], acc[j][q]);
            };
            constexpr int GROUPS = TWD / K, TAIL = TWD % K;
            const char* xp = xr + (K - 1) * 64;
            const char* dp = dr;
#pragma unroll 1
            for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    step(xp + t * 64, dp + t * 64, t);
                    if (t == K / 2) asm volatile("" ::: "memory");      // keeps hipcc from hoisting all 2 K ds_reads of the group (register pressure)
                }
                xp += K * 64;
                dp += K * 64;
            }
#pragma unroll
            for (int t = 0; t < TAIL; ++t) step(xp + t * 64, dp + t * 64, t);
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_wg);      // [TH][NT][32]
    if (worker) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float* dst = red + ((r * NT + ky * K + j) * 32 + cg * 8);
            *reinterpret_cast<float4*>(dst) = make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[j][2].x, acc[j][2].y, acc[j][3].x, acc[j][3].y);
        }
        if (ky == 0) {
            float* dst = red + ((r * NT + K * K) * 32 + cg * 8);
            *reinterpret_cast<float4*>(dst) = make_float4(accb[0].x, accb[0].y, accb[1].x, accb[1].y);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(accb[2].x, accb[2].y, accb[3].x, accb[3].y);
        }
    }
    __syncthreads();
    float* out = partials + (int64_t)blockIdx.x * NT * C;
    for (int i = tid; i < NT * 32; i += 256) {
        const int tap = i >> 5, c = i & 31;
        float s = 0.f;
        for (int rr = 0; rr < TH; ++rr) s += red[(rr * NT + tap) * 32 + c];
        out[(int64_t)tap * C + c0 + c] = s;
    }
}

struct BwDmaGeom {
    int ok, th, twd, tiles_h, tiles_w, slabs, bx;
    size_t lds_bytes;
};

static int use_bw_dma() { static int v = env_int("ISEG_DW_BW_DMA", 1); return v != 0; }

static BwDmaGeom bw_dma_geom(int N, int H, int W, int C, int K, int dil, size_t elem) {
    BwDmaGeom g;
    g.ok = 0;
    if (!use_bw_dma() || elem != 2 || K != 7 || dil != 1 || C % 32 != 0) return g;
    g.twd = W <= 16 ? 16 : 32;
    const int t8 = (H + 7) / 8, t9 = (H + 8) / 9;
    g.th = t9 < t8 ? 9 : 8;
    if (g.th > H) g.th = H;
    g.tiles_h = (H + g.th - 1) / g.th;
    g.tiles_w = (W + g.twd - 1) / g.twd;
    g.slabs = C / 32;
    const int64_t ntiles = (int64_t)N * g.tiles_h * g.tiles_w;
    if (ntiles >= (1ll << 30)) return g;
    // two resident workgroups per CU: ~512 in flight; every workgroup of a slab gets the same number of tiles (+-1)
    static const int slots = env_int("ISEG_DW_BW_DMA_SLOTS", 512);
    int64_t cap = slots / g.slabs;
    if (cap < 8) cap = 8;
    const int64_t rounds = ceil_div64(ntiles, cap);
    int64_t bx = ceil_div64(ntiles, rounds);
    if (bx % 8 && (bx + 7) / 8 * 8 <= ntiles) bx = (bx + 7) / 8 * 8;
    g.bx = (int)bx;
    const int ih = g.th + K - 1;
    const size_t pieces = (size_t)(ih * (g.twd + K) + 15) / 16 + (size_t)(g.th * (g.twd + 1) + 15) / 16;
    const size_t red = (size_t)g.th * (K * K + 1) * 32 * sizeof(float);
    g.lds_bytes = pieces * 1024 > red ? pieces * 1024 : red;
    g.ok = 1;
    return g;
}

static void launch_bw_dma(const void* x, const void* dy, float* ws, int N, int H, int W, int C, int pad_t, int pad_l, const BwDmaGeom& g,
                          hipStream_t s) {
    if (g.twd == 16)
        hipLaunchKernelGGL((dwconv_bwd_weight_dma_kernel<7, 16>), dim3(g.bx, g.slabs), dim3(256), g.lds_bytes, s, (const bf16_t*)x,
                           (const bf16_t*)dy, ws, N, H, W, C, pad_t, pad_l, g.th, g.tiles_h, g.tiles_w);
    else
        hipLaunchKernelGGL((dwconv_bwd_weight_dma_kernel<7, 32>), dim3(g.bx, g.slabs), dim3(256), g.lds_bytes, s, (const bf16_t*)x,
                           (const bf16_t*)dy, ws, N, H, W, C, pad_t, pad_l, g.th, g.tiles_h, g.tiles_w);
}

// DMA-tiled bf16 forward / data-gradient (K x K, dil == 1, C % 32 == 0): the ConvNeXt stages' variant.
//   * persistent workgroups walk (channel slab, image, row band, column band) units; the input tile with its halo goes HBM -> LDS by
//     global_load_lds_dwordx4 exactly as in the weight-gradient kernel above (zero page for the padding), two workgroups per CU;
//   * lane (cg, row, seg): 8 channels, output row `row` of a 16-row band, TWO consecutive output pixels (seg = the wavefront, so a 16-lane
//     LDS read group spans four rows: the 39 / 23-pixel row stride puts them on distinct banks);
//   * per kernel row: the K x 8 fp32 weights of the lane's channels (LDS, broadcast reads) and TWO + K - 1 input pixels, each unpacked once
//     and fed to up to K taps: (TWO + K - 1) x 8 unpack + TWO x K x 4 packed FMAs;
//   * bias, residual add (`add`) and the bf16 rounding happen on the accumulators; the weights are re-staged only when the slab changes.
template <int K, int TWO, int TH>
__global__ __launch_bounds__(256, 2) void dwconv_fwd_dma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const bf16_t* __restrict__ add,
                                                                bf16_t* __restrict__ y, int N, int H, int W, int C, int pad_t, int pad_l,
                                                                int flip, int tiles_h, int tiles_w, int slabs) {
    constexpr int SEGS = 64 / TH, TWD = SEGS * TWO, IH = TH + K - 1, IW = TWD + K - 1, IWP = IW + 1;      // 4 channel groups x TH rows x SEGS segments
    static_assert((IWP * 64) % 256 == 192, "row stride must spread four consecutive rows over the banks");
    constexpr int XPIECES = (IH * IWP + 15) / 16, MAXPX = (XPIECES + 3) / 4;
    extern __shared__ __attribute__((aligned(1024))) char smem_fd[];
    char* xt = smem_fd;                                                   // [IH][IWP][32] bf16
    float* wl = reinterpret_cast<float*>(smem_fd + XPIECES * 1024);       // [K*K][32] fp32
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = tid & 3, row = (tid >> 2) % TH, seg = tid / (4 * TH);
    int lb = blockIdx.x;
    if (gridDim.x % 8 == 0) lb = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    const int ntiles = N * tiles_h * tiles_w;
    const int units = ntiles * slabs;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(dw_zero_page) + (lane & 3) * 8;
    int xrc[MAXPX];
#pragma unroll
    for (int i = 0; i < MAXPX; ++i) {
        const int pix = (wid + 4 * i) * 16 + (lane >> 2);
        const int rr = pix / IWP, cc = pix - rr * IWP;
        xrc[i] = ((cc >= IW || rr >= IH ? 0x7fff : rr) << 16) | cc;
    }
    int cur_slab = -1;
    f32x2 bv[4];
    for (int u = lb; u < units; u += gridDim.x) {
        // tile-major, slab-minor (round 6): the channel slabs of one pixel tile are consecutive units, i.e. neighbouring workgroups of ONE XCD (lb is
        // XCD-contiguous) running at the same time -- a 128-byte line holds two 64-byte slab pieces of a pixel, and with the slab-major order of
        // rounds 3-5 its second half was fetched again by another XCD much later (PMC: 59.9 MB per launch for 39 MB algorithmic)
        int b = u / slabs;
        const int slab = u - b * slabs;
        const int tw_i = b % tiles_w;
        b /= tiles_w;
        const int th_i = b % tiles_h;
        const int n = b / tiles_h;
        const int h0 = th_i * TH, w0 = tw_i * TWD;
        const int c0 = slab * 32;
        __syncthreads();      // previous unit consumed
        const bf16_t* xb = x + (((n * H + h0 - pad_t) * W + w0 - pad_l) * C + c0 + (lane & 3) * 8);
#pragma unroll
        for (int i = 0; i < MAXPX; ++i) {
            const int p = wid + 4 * i;
            if (p < XPIECES) {
                int rc = xrc[i];
                asm volatile("" : "+v"(rc));
                const int rr = rc >> 16, cc = rc & 0xffff;
                const bool ok = (unsigned)(h0 - pad_t + rr) < (unsigned)H && (unsigned)(w0 - pad_l + cc) < (unsigned)W;
                const bf16_t* src = ok ? xb + __mul24(__mul24(rr, W) + cc, C) : zero;
#ifndef DWF_ABL_NODMA      // (ablation builds, tools/ab_build.py: no tile fill / no taps / no stores)
                __builtin_amdgcn_global_load_lds((dw_glb_ptr)src, (dw_lds_ptr)(xt + p * 1024), 16, 0, 0);
#else
                asm volatile("" ::"v"(src));
#endif
            }
        }
        // (behind the tile's DMA, so the two round trips overlap: staged in front of it, the weights cost a dependent trip of their own per slab)
        if (slab != cur_slab) {      // workgroup-uniform
            cur_slab = slab;
            for (int i = tid; i < K * K * 8; i += 256) {      // float4 granules: [tap][32]
                int tap = i >> 3;
                const int c4 = (i & 7) * 4;
                if (flip) tap = K * K - 1 - tap;
                reinterpret_cast<float4*>(wl)[i] = *reinterpret_cast<const float4*>(w + tap * C + c0 + c4);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = bias ? f32x2{bias[c0 + cg * 8 + 2 * q], bias[c0 + cg * 8 + 2 * q + 1]} : f32x2{0.f, 0.f};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x2 acc[TWO][4];
#pragma unroll
        for (int t = 0; t < TWO; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][q] = bv[q];
#ifdef DWF_ABL_NOTAPS
        constexpr int KY_END = 1;
#else
        constexpr int KY_END = K;
#endif
#pragma unroll 1
        for (int ky = 0; ky < KY_END; ++ky) {
            f32x2 wr[K][4];
            const float* wrow = wl + (ky * K) * 32 + cg * 8;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float4 a = *reinterpret_cast<const float4*>(wrow + kx * 32), c = *reinterpret_cast<const float4*>(wrow + kx * 32 + 4);
                wr[kx][0] = f32x2{a.x, a.y};
                wr[kx][1] = f32x2{a.z, a.w};
                wr[kx][2] = f32x2{c.x, c.y};
                wr[kx][3] = f32x2{c.z, c.w};
            }
            const char* xr = xt + ((row + ky) * IWP + seg * TWO) * 64 + cg * 16;
#pragma unroll
            for (int s = 0; s < TWO + K - 1; ++s) {
                f32x2 xs[4];
                unpack_bf16x8(xr + s * 64, xs);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int t = s - kx;
                    if (t >= 0 && t < TWO) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[t][q] = __builtin_elementwise_fma(xs[q], wr[kx][q], acc[t][q]);
                    }
                }
            }
        }
        const int oh = h0 + row;
        if (oh < H) {
            const int ow0 = w0 + seg * TWO;
            const int offr = (n * H + oh) * W * C + c0 + cg * 8;
            const int off0 = offr + ow0 * C;
            if (add) {
                // every load is issued, clamped to the row's last pixel (a segment may start past the edge of a narrow plane): a branch per
                // load makes hipcc wait for each
                f32x2 av[TWO][4];
#pragma unroll
                for (int t = 0; t < TWO; ++t) unpack_bf16x8(reinterpret_cast<const char*>(add + offr + (ow0 + t < W ? ow0 + t : W - 1) * C), av[t]);
#pragma unroll
                for (int t = 0; t < TWO; ++t)
                    if (ow0 + t < W) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[t][q] += av[t][q];
                    }
            }
#pragma unroll
            for (int t = 0; t < TWO; ++t)
#ifdef DWF_ABL_NOOUT
                if (ow0 + t < W && acc[t][0].x == 1234.5f) {
#else
                if (ow0 + t < W) {
#endif
                    bf16x8 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[2 * q] = (bf16_t)acc[t][q].x;
                        o[2 * q + 1] = (bf16_t)acc[t][q].y;
                    }
                    *reinterpret_cast<bf16x8*>(y + off0 + t * C) = o;
                }
        }
    }
}

static int use_fwd_dma() { static int v = env_int("ISEG_DW_FWD_DMA", 1); return v != 0; }

static bool launch_fwd_dma(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int K,
                           int dil, int pad_t, int pad_l, int flip, hipStream_t s) {
    if (!use_fwd_dma() || K != 7 || dil != 1 || C % 32 != 0) return false;
    // tile shapes (rows x columns, output pixels per lane): 16 x 32 (8) for the wide planes, 8 x 32 (4) when that leaves fewer than two tiles per
    // resident workgroup (32 x 32 planes: three smaller workgroups per CU overlap each other's fills), 16 x 16 (4) for 16-pixel planes.
    // Measured (16 images, us, LDS kernel -> this one): 128x128x96 70.4 -> 53.8, 64x64x192 37.2 -> 33.9, 16x16x768 14.5 -> 12.1; 32x32x384 with
    // the 16 x 32 tile 19.2 -> 21.5 (one tile per workgroup, nothing overlaps the fill), with the 8 x 32 tile -> 17.5.
    // A double-buffered form (ONE workgroup per CU, the next tile's DMA issued from inline assembly so that hipcc puts no vmcnt(0) in front of
    // the LDS reads) measured 65 us at 128x128x96 against 53 us for two single-buffered workgroups per CU: with one wavefront per SIMD nothing
    // hides the LDS-read and FMA latencies
    const int slabs = C / 32;
    int two = W <= 16 ? 4 : 8, th = 16;
    if (two == 8 && (int64_t)N * ((H + 15) / 16) * ((W + 31) / 32) * slabs < 1536) {      // (64x64x192: 33.6 -> 32.3 us; 128x128x96 would lose: 53.0 -> 58.5)
        two = 4;
        th = 8;
    }
    static const int force_small = env_int("ISEG_DW_FWD_SMALL", 0);      // experiment: the 8 x 32 (4) tile everywhere
    if (force_small && W > 16) {
        two = 4;
        th = 8;
    }
    const int twd = (64 / th) * two;
    const int tiles_h = (H + th - 1) / th, tiles_w = (W + twd - 1) / twd;
    const int64_t units = (int64_t)N * tiles_h * tiles_w * slabs;
    if (units >= (1ll << 30)) return false;
    // resident workgroups: two per CU for the wide variant, three for the others; every workgroup gets the same number of units (+-1)
    const int64_t slots = two == 8 ? 512 : 768;
    const int64_t rounds = ceil_div64(units, slots);
    int64_t nwg = ceil_div64(units, rounds);
    if (nwg % 8 && (nwg + 7) / 8 * 8 <= units && (nwg + 7) / 8 * 8 <= slots) nwg = (nwg + 7) / 8 * 8;
    const int ih = th + K - 1, iwp = twd + K;
    const size_t lds = (size_t)((ih * iwp + 15) / 16) * 1024 + (size_t)K * K * 32 * sizeof(float);
#define DW_FWD_DMA(TWO_, TH_)                                                                                                                  \
    hipLaunchKernelGGL((dwconv_fwd_dma_kernel<7, TWO_, TH_>), dim3((unsigned)nwg), dim3(256), lds, s, (const bf16_t*)x, w, bias, (const bf16_t*)add, \
                       (bf16_t*)y, N, H, W, C, pad_t, pad_l, flip, tiles_h, tiles_w, slabs)
    if (two == 8) DW_FWD_DMA(8, 16);
    else if (th == 8) DW_FWD_DMA(4, 8);
    else DW_FWD_DMA(4, 16);
#undef DW_FWD_DMA
    return true;
}

struct BwGeom {
    int cv, gs, rt, slabs, wseg, ipl, bx;
    int lds, tiles_h, tiles_w;   // LDS-tiled variant
    size_t lds_bytes;
};

static int use_bw_lds() { static int v = env_int("ISEG_DW_BW_LDS", 1); return v != 0; }
static int bw_lds_min_w() { static int v = env_int("ISEG_DW_BW_LDS_MINW", 48); return v; }

static BwGeom bw_geom(int N, int H, int W, int C, int K, int dil, size_t elem) {
    BwGeom g;
    g.lds = 0;
    // bf16 takes the register-batched kernel below at every plane size (measured at 64x64x192: 69 us vs 88 us for the LDS tiles, equal
    // at 128x128x96); the LDS-tiled variant serves fp32 storage
    static const int lds_bf16 = env_int("ISEG_DW_BW_LDS_BF16", 0);
    if (use_bw_lds() && dil == 1 && C % 8 == 0 && W >= bw_lds_min_w() && (elem != 2 || lds_bf16)) {
        // channel slab of <= 6 groups (48 channels): 6 x K x rt lanes.  Small planes (W <= 32: one tile spans the row) take
        // the widest slab whose lane rows still cover the whole image height, so one tile = one image plane.
        static const int max_groups = env_int("ISEG_DW_BW_LDS_GROUPS", 3);   // measured 1..6 at 128x128x96 / 64x64x192: 163/107, 155/101, 137/88, 141/99, 144/92 us
        int gs = groups_per_slab(C, 8, max_groups);
        if (W <= BWW) {
            gs = 1;
            for (int cand = 6; cand >= 1; --cand)
                if ((C / 8) % cand == 0 && 256 / (cand * K) >= H) {
                    gs = cand;
                    break;
                }
        }
        int rt = 256 / (gs * K);
        if (rt > H) rt = H;
        const size_t tile_bytes = ((size_t)(rt + K - 1) * (BWW + K - 1) + (size_t)rt * BWW) * gs * 8 * elem;
        const size_t slab_bytes = (size_t)rt * (K * K + 1) * gs * 8 * 4;      // the per-lane sum slabs reuse the tile area
        const size_t bytes = tile_bytes > slab_bytes ? tile_bytes : slab_bytes;
        if (rt >= 1 && bytes <= 80 * 1024) {
            g.lds = 1;
            g.cv = 8;
            g.gs = gs;
            g.rt = rt;
            g.slabs = (C / 8) / gs;
            g.tiles_h = (H + rt - 1) / rt;
            g.tiles_w = (W + BWW - 1) / BWW;
            const int64_t ntiles = (int64_t)N * g.tiles_h * g.tiles_w;
            int64_t bx = 512 / g.slabs;
            if (bx < 32) bx = 32;
            if (bx > ntiles) bx = ntiles;
            g.bx = (int)bx;
            g.lds_bytes = bytes;
            g.wseg = g.ipl = 0;
            return g;
        }
    }
    g.cv = (C % 8 == 0 && bw_cv() == 8) ? 8 : 4;
    g.gs = groups_per_slab(C, g.cv, g.cv == 8 ? 16 : 12);
    g.rt = 256 / (g.gs * K);
    if (g.rt < 1) g.rt = 1;
    g.slabs = (C / g.cv) / g.gs;
    g.wseg = W <= 32 ? W : 32;
    const int nseg = (W + g.wseg - 1) / g.wseg;
    const int64_t items = (int64_t)N * H * nseg;
    // aim at ~1024 blocks in total: enough waves to hide L2 latency, small enough partial buffers.  The register-batched bf16 path
    // (W <= 32) keeps ~70 loads in flight per lane by itself and runs 2 waves per SIMD: ~512 blocks = one resident round
    const bool batched = elem == 2 && g.cv == 4 && dil == 1 && g.wseg <= BWSEG;
    int64_t target = (batched ? env_int("ISEG_DW_BW_BLOCKS", 512) : 1024) / g.slabs;
    if (target < (batched ? 8 : 64)) target = batched ? 8 : 64;
    int64_t ipl = ceil_div64(items, (int64_t)g.rt * target);
    if (ipl < 1) ipl = 1;
    g.ipl = (int)ipl;
    g.bx = (int)ceil_div64(items, (int64_t)g.rt * g.ipl);
    if (g.bx < 1) g.bx = 1;
    return g;
}

template <class T, int K, int CV>
int launch_fwd(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int dil,
               int pad_t, int pad_l, int flip, hipStream_t s) {
    const int gs = groups_per_slab(C, CV, 16);
    const int pt = 256 / gs;
    const int slabs = (C / CV) / gs;
    const int64_t tiles = (int64_t)N * H * ((W + TW - 1) / TW);
    int64_t bx = ceil_div64(tiles, pt);
    const int64_t cap = 256 * 8 / slabs > 1 ? 256 * 8 / slabs : 1;
    if (bx > cap) bx = cap;
    const size_t lds = (size_t)K * K * gs * CV * sizeof(float);
#define DW_LAUNCH(D1, RL)                                                                                                         \
    hipLaunchKernelGGL((dwconv_fwd_kernel<T, K, CV, D1, RL>), dim3((unsigned)bx, slabs), dim3(256), lds, s, (const T*)x, w, bias, \
                       (const T*)add, (T*)y, N, H, W, C, dil, pad_t, pad_l, flip, gs, pt)
    if (dil == 1) {
        if (fwd_rolled()) DW_LAUNCH(true, true);
        else DW_LAUNCH(true, false);
    } else {
        DW_LAUNCH(false, true);
    }
#undef DW_LAUNCH
    return iseg_check_launch("iseg_dwconv2d");
}

static int use_lds() { static int v = env_int("ISEG_DW_LDS", 1); return v != 0; }
// channel groups (of 8) per workgroup slab; 0 = automatic.  Measured at the four ConvNeXt-T stages (16 images, us per launch):
//   groups      2      3      4      6      8
//   128x128x96  79.8   89.8   69.8   79.8   79.9
//   64x64x192   36.1   53.6   39.4   49.8   49.2
//   32x32x384   19.0   29.8   20.5   30.8   28.9
//   16x16x768   14.3   14.4   14.9   15.9   15.8
// Power-of-two slabs give square-ish pixel tiles (32x16 / 16x16: halo amplification 1.6 / 1.9 instead of 2.2 at 10x16) and keep
// all 256 lanes busy; the widest planes prefer 64-byte pixel pieces (4 groups) for their L2 -> L1 line use.
static int fwd_lds_groups() { static int v = env_int("ISEG_DW_LDS_GROUPS", 0); return v < 0 ? 0 : (v > 16 ? 16 : v); }

template <class T, int K>
bool launch_fwd_lds(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int dil,
                    int pad_t, int pad_l, int flip, hipStream_t s) {
    const int want = fwd_lds_groups() ? fwd_lds_groups() : ((int64_t)H * W >= 128 * 128 ? 4 : 2);
    const int gs = groups_per_slab(C, 8, want);
    const int TH = (256 / gs) / (TWB / TW);
    if (TH < 1) return false;
    const int halo = (K - 1) * dil;
    const size_t lds = (size_t)K * K * gs * 8 * sizeof(float) + (size_t)(TH + halo) * (TWB + halo) * (gs * 8 * sizeof(T) + 16);
    if (lds > 80 * 1024) return false;
    const int slabs = (C / 8) / gs;
    const int tiles_h = (H + TH - 1) / TH, tiles_w = (W + TWB - 1) / TWB;
    const int64_t bx = (int64_t)N * tiles_h * tiles_w;
    if (bx >= (1ll << 31)) return false;
    if (dil == 1)
        hipLaunchKernelGGL((dwconv_fwd_lds_kernel<T, K, true>), dim3((unsigned)bx, slabs), dim3(256), lds, s, (const T*)x, w, bias,
                           (const T*)add, (T*)y, N, H, W, C, dil, pad_t, pad_l, flip, gs, TH, tiles_h, tiles_w);
    else
        hipLaunchKernelGGL((dwconv_fwd_lds_kernel<T, K, false>), dim3((unsigned)bx, slabs), dim3(256), lds, s, (const T*)x, w, bias,
                           (const T*)add, (T*)y, N, H, W, C, dil, pad_t, pad_l, flip, gs, TH, tiles_h, tiles_w);
    return true;
}

template <class T, int K>
int launch_fwd_cv(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int dil,
                  int pad_t, int pad_l, int flip, hipStream_t s) {
    if (use_lds() && C % 8 == 0 && launch_fwd_lds<T, K>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, s))
        return iseg_check_launch("iseg_dwconv2d");
    if (C % 8 == 0 && fwd_cv() == 8) return launch_fwd<T, K, 8>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, s);
    return launch_fwd<T, K, 4>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, s);
}

template <class T, int K, int CV>
void launch_bw(const void* x, const void* dy, float* ws, int N, int H, int W, int C, int dil, int pad_t, int pad_l, const BwGeom& g,
               hipStream_t s) {
    const size_t lds = (size_t)g.rt * (K * K + 1) * g.gs * CV * sizeof(float);
    if (dil == 1)
        hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, K, CV, true>), dim3(g.bx, g.slabs), dim3(256), lds, s, (const T*)x,
                           (const T*)dy, ws, N, H, W, C, dil, pad_t, pad_l, g.gs, g.rt, g.wseg, g.ipl);
    else
        hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, K, CV, false>), dim3(g.bx, g.slabs), dim3(256), lds, s, (const T*)x,
                           (const T*)dy, ws, N, H, W, C, dil, pad_t, pad_l, g.gs, g.rt, g.wseg, g.ipl);
}

template <class T, int K>
void launch_bw_cv(const void* x, const void* dy, float* ws, int N, int H, int W, int C, int dil, int pad_t, int pad_l, const BwGeom& g,
                  hipStream_t s) {
    if (g.cv == 8) launch_bw<T, K, 8>(x, dy, ws, N, H, W, C, dil, pad_t, pad_l, g, s);
    else launch_bw<T, K, 4>(x, dy, ws, N, H, W, C, dil, pad_t, pad_l, g, s);
}

}  // namespace

bool iseg_dwconv7_mfma_launch(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int K, int dil,
                              int pad_t, int pad_l, int flip, hipStream_t s);      // dwconv_mfma.hip

extern "C" int iseg_dwconv2d_fwd(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W,
                                 int C, int K, int dil, int pad_t, int pad_l, int flip, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && w && y, "iseg_dwconv2d_fwd: null pointer");
    ISEG_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "iseg_dwconv2d_fwd: C=%d must be a multiple of 8", C);
    ISEG_REQUIRE(K == 3 || K == 5 || K == 7, "iseg_dwconv2d_fwd: kernel size %d unsupported (3,5,7)", K);
    ISEG_REQUIRE(dil >= 1, "iseg_dwconv2d_fwd: dilation must be >= 1");
    ISEG_REQUIRE((int64_t)N * H * W * C < (1ll << 31), "iseg_dwconv2d_fwd: more than 2^31 elements");
#define DW_FWD(T)                                                                                             \
    (K == 7   ? launch_fwd_cv<T, 7>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, stream)          \
     : K == 5 ? launch_fwd_cv<T, 5>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, stream)          \
              : launch_fwd_cv<T, 3>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, stream))
    // 7 x 7, bf16, C % 32 == 0: the banded products on the matrix cores (dwconv_mfma.hip, round 5; ISEG_DW_MFMA=0 keeps the VALU kernels)
    if (dtype == ISEG_BF16 && iseg_dwconv7_mfma_launch(x, w, bias, add, y, N, H, W, C, K, dil, pad_t, pad_l, flip, stream))
        return iseg_check_launch("iseg_dwconv2d (mfma)");
    if (dtype == ISEG_BF16 && launch_fwd_dma(x, w, bias, add, y, N, H, W, C, K, dil, pad_t, pad_l, flip, stream))
        return iseg_check_launch("iseg_dwconv2d");
    return dtype == ISEG_BF16 ? DW_FWD(bf16_t) : DW_FWD(float);
#undef DW_FWD
}

extern "C" size_t iseg_dwconv2d_bwd_weight_workspace_bytes(int N, int H, int W, int C, int K) {
    // upper bound over storage dtypes and over the LDS / non-LDS variants
    const BwGeom g = bw_geom(N, H, W, C, K, 1, 4);
    const BwGeom g2 = bw_geom(N, H, W, C, K, 2, 4);
    const BwGeom g3 = bw_geom(N, H, W, C, K, 1, 2);
    int bx = g.bx > g2.bx ? g.bx : g2.bx;
    if (g3.bx > bx) bx = g3.bx;
    const BwDmaGeom gd = bw_dma_geom(N, H, W, C, K, 1, 2);
    if (gd.ok && gd.bx > bx) bx = gd.bx;
    return (size_t)bx * (K * K + 1) * C * sizeof(float);
}

extern "C" int iseg_dwconv2d_bwd_weight(const void* x, const void* dy, float* dw, float* db, int accumulate, int N, int H, int W,
                                        int C, int K, int dil, int pad_t, int pad_l, int dtype, void* ws, size_t ws_bytes,
                                        hipStream_t stream) {
    ISEG_REQUIRE(x && dy && dw, "iseg_dwconv2d_bwd_weight: null pointer");
    ISEG_REQUIRE(C % 8 == 0, "iseg_dwconv2d_bwd_weight: C=%d must be a multiple of 8", C);
    ISEG_REQUIRE(K == 3 || K == 5 || K == 7, "iseg_dwconv2d_bwd_weight: kernel size %d unsupported", K);
    ISEG_REQUIRE((int64_t)N * H * W * C < (1ll << 31), "iseg_dwconv2d_bwd_weight: more than 2^31 elements");
    const BwDmaGeom gd = bw_dma_geom(N, H, W, C, K, dil, dtype == ISEG_BF16 ? 2 : 4);
    BwGeom g = bw_geom(N, H, W, C, K, dil, dtype == ISEG_BF16 ? 2 : 4);
    ISEG_REQUIRE(g.gs * K * g.rt <= 256, "iseg_dwconv2d_bwd_weight: slab does not fit a block");
    if (gd.ok) g.bx = gd.bx;
    const size_t need = (size_t)g.bx * (K * K + 1) * C * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_dwconv2d_bwd_weight: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    float* const arena = iseg_deferred_partials(need, dw, db, accumulate, stream);      // (see common.h: deferred reductions)
    if (arena) ws = arena;
    if (gd.ok) {
        launch_bw_dma(x, dy, (float*)ws, N, H, W, C, pad_t, pad_l, gd, stream);
    } else if (g.lds) {
#define DW_BWL(T, KK)                                                                                                               \
    hipLaunchKernelGGL((dwconv_bwd_weight_lds_kernel<T, KK>), dim3(g.bx, g.slabs), dim3(256), g.lds_bytes, stream, (const T*)x,         \
                       (const T*)dy, (float*)ws, N, H, W, C, pad_t, pad_l, g.gs, g.rt, g.tiles_h, g.tiles_w)
        if (dtype == ISEG_BF16) {
            if (K == 7) DW_BWL(bf16_t, 7);
            else if (K == 5) DW_BWL(bf16_t, 5);
            else DW_BWL(bf16_t, 3);
        } else {
            if (K == 7) DW_BWL(float, 7);
            else if (K == 5) DW_BWL(float, 5);
            else DW_BWL(float, 3);
        }
#undef DW_BWL
    } else if (dtype == ISEG_BF16) {
        if (K == 7) launch_bw_cv<bf16_t, 7>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
        else if (K == 5) launch_bw_cv<bf16_t, 5>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
        else launch_bw_cv<bf16_t, 3>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
    } else {
        if (K == 7) launch_bw_cv<float, 7>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
        else if (K == 5) launch_bw_cv<float, 5>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
        else launch_bw_cv<float, 3>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
    }
    const int n = (K * K + 1) * C;
    if (arena) iseg_deferred_push((const float*)ws, g.bx, n, n, dw, db, K * K * C, 1.f, stream);
    else launch_reduce_rows((const float*)ws, g.bx, n, 0, 1, n, dw, db, K * K * C, 0, 1.f, accumulate, stream);
    return iseg_check_launch("iseg_dwconv2d_bwd_weight");
}
