// Depthwise KxK convolution on NHWC tensors (keras.layers.DepthwiseConv2D, padding="same", stride 1,
// dilation d) -- the 7x7 depthwise of backbones/convnext.py:25,50 (dilated by build_dilated_convnext :245-266)
// and the 3x3 depthwise of layers/dcn_v3/dcn_v3.py.  49 FMA per element but only 4 B of HBM traffic: the kernels
// are limited by how many L1/L2 loads they keep in flight, so:
//   * a lane owns 8 consecutive channels (one 16-B load per pixel) and TW output pixels along W, and slides a
//     register window over the input row so each loaded pixel feeds up to KW taps;
//   * every load is issued unconditionally from a clamped address and zeroed by a select afterwards (no divergent
//     branch between loads, so a whole kernel row's loads are in flight together); interior tiles skip the selects;
//   * weights of the block's channel slab sit in LDS as fp32 (the fp32 master kernel is read directly);
//   * backward-data is the same kernel with the taps flipped and the complementary padding, and can add the
//     residual branch's gradient on the way out (dx = dres + dwconv^T(dy));
//   * backward-weight gives each lane one kernel row (KW taps x 8 channels of accumulators) and a run of
//     (image row, W-segment) items, slides a KW-wide register window along W, then reduces lanes -> block (LDS) ->
//     grid (fixed-order partial sums, deterministic).
#include "common.h"
#include "iseg_hip.h"

namespace {

constexpr int TW = 4;

static inline int groups_per_slab(int C) {
    const int G = C / 8;
    int best = 1;
    for (int g = 1; g <= 16 && g <= G; ++g)
        if (G % g == 0) best = g;
    return best;
}

// unconditional load from a clamped pixel index, zeroed when the true index is outside [0, W)
template <class T, bool CHECK>
__device__ __forceinline__ void load_px(const T* __restrict__ row, int iw, int W, int C, bool row_ok, float* out) {
    if (CHECK) {
        const bool ok = row_ok && (unsigned)iw < (unsigned)W;
        const int iwc = min(max(iw, 0), W - 1);
        load8<T>(row + (int64_t)iwc * C, out);
#pragma unroll
        for (int u = 0; u < 8; ++u) out[u] = ok ? out[u] : 0.f;
    } else {
        load8<T>(row + (int64_t)iw * C, out);
    }
}

template <class T, int K, bool DIL1, bool CHECK>
__device__ __forceinline__ void dw_accumulate_row(const T* __restrict__ xr, const float* __restrict__ wrow, int w0, int pad_l, int W,
                                                  int C, int dil, int sc, float (&acc)[TW][8]) {
    if (DIL1) {
        float xin[TW + K - 1][8];
#pragma unroll
        for (int s = 0; s < TW + K - 1; ++s) load_px<T, CHECK>(xr, w0 - pad_l + s, W, C, true, xin[s]);
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
            float xv[TW][8];
#pragma unroll
            for (int t = 0; t < TW; ++t) load_px<T, true>(xr, w0 + t + kw * dil - pad_l, W, C, true, xv[t]);
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[t][u] = fmaf(xv[t][u], wv[u], acc[t][u]);
        }
    }
}

template <class T, int K, bool DIL1>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const T* __restrict__ add,
                                                         T* __restrict__ y, int N, int H, int W, int C, int dil, int pad_t,
                                                         int pad_l, int flip, int gs, int pt) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [K*K][gs*8]
    const int slab_c0 = blockIdx.y * gs * 8;
    const int sc = gs * 8;
    for (int i = threadIdx.x; i < K * K * sc; i += blockDim.x) {
        int tap = i / sc;
        const int c = i % sc;
        if (flip) tap = K * K - 1 - tap;
        wl[i] = w[(int64_t)tap * C + slab_c0 + c];
    }
    __syncthreads();
    const int cg = threadIdx.x % gs, ptile = threadIdx.x / gs;
    if (ptile >= pt) return;
    const int wtiles = (W + TW - 1) / TW;
    const int tiles_total = N * H * wtiles;
    const int c0 = slab_c0 + cg * 8;
    float bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) bv[u] = bias ? bias[c0 + u] : 0.f;

    for (int tile = blockIdx.x * pt + ptile; tile < tiles_total; tile += gridDim.x * pt) {
        const int wt = tile % wtiles;
        const int nh = tile / wtiles;
        const int h = nh % H;
        const int n = nh / H;
        const int w0 = wt * TW;
        float acc[TW][8];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[t][u] = bv[u];
        const T* xn = x + (int64_t)n * H * W * C + c0;
        const bool interior = DIL1 && (w0 - pad_l >= 0) && (w0 - pad_l + TW + K - 2 < W);
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int ih = h + kh * dil - pad_t;
            if ((unsigned)ih >= (unsigned)H) continue;
            const T* xr = xn + (int64_t)ih * W * C;
            const float* wrow = wl + kh * K * sc + cg * 8;
            if (interior) dw_accumulate_row<T, K, DIL1, false>(xr, wrow, w0, pad_l, W, C, dil, sc, acc);
            else dw_accumulate_row<T, K, DIL1, true>(xr, wrow, w0, pad_l, W, C, dil, sc, acc);
        }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int ow = w0 + t;
            if (ow < W) {
                const int64_t off = (((int64_t)n * H + h) * W + ow) * C + c0;
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
}

// dw[kh][kw][c] = sum_{n,h,w} x[n, h+kh*d-pt, w+kw*d-pl, c] * dy[n,h,w,c];  db[c] = sum dy
// thread = (channel group, kernel row kh, item lane); an item is (image row, W segment of `wseg` pixels).
template <class T, int K, bool DIL1>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                float* __restrict__ partials, int N, int H, int W, int C,
                                                                int dil, int pad_t, int pad_l, int gs, int rt, int wseg, int ipl) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [(K*K+1)][gs*8]
    const int sc = gs * 8;
    const int slab_c0 = blockIdx.y * sc;
    const int nred = (K * K + 1) * sc;
    for (int i = threadIdx.x; i < nred; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    const int cg = threadIdx.x % gs;
    const int kh = (threadIdx.x / gs) % K;
    const int rl = threadIdx.x / (gs * K);
    if (rl < rt) {
        const int c0 = slab_c0 + cg * 8;
        float acc[K][8], accb[8];
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[j][u] = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) accb[u] = 0.f;
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
            if (DIL1) {
                float win[K][8];
#pragma unroll
                for (int j = 0; j < K - 1; ++j) load_px<T, true>(xr, ws - pad_l + j, W, C, row_ok, win[j]);
                for (int base = ws; base < we; base += K) {
#pragma unroll
                    for (int t = 0; t < K; ++t) {
                        const int ow = base + t;
                        load_px<T, true>(xr, ow - pad_l + (K - 1), W, C, row_ok, win[(t + K - 1) % K]);
                        float d[8];
                        load_px<T, true>(dyr, ow < we ? ow : -1, W, C, true, d);
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
            } else {
                for (int ow = ws; ow < we; ++ow) {
                    float d[8];
                    load8<T>(dyr + (int64_t)ow * C, d);
                    if (kh == 0) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) accb[u] += d[u];
                    }
                    float xv[K][8];
#pragma unroll
                    for (int j = 0; j < K; ++j) load_px<T, true>(xr, ow + j * dil - pad_l, W, C, row_ok, xv[j]);
#pragma unroll
                    for (int j = 0; j < K; ++j)
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc[j][u] = fmaf(xv[j][u], d[u], acc[j][u]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int u = 0; u < 8; ++u) atomicAdd(&red[(kh * K + j) * sc + cg * 8 + u], acc[j][u]);
        if (kh == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) atomicAdd(&red[K * K * sc + cg * 8 + u], accb[u]);
        }
    }
    __syncthreads();
    // partial layout: [block.x][(K*K+1)][C]
    float* out = partials + (int64_t)blockIdx.x * (K * K + 1) * C;
    for (int i = threadIdx.x; i < nred; i += blockDim.x) {
        const int tap = i / sc, c = i % sc;
        out[(int64_t)tap * C + slab_c0 + c] = red[i];
    }
}

struct BwGeom {
    int gs, rt, slabs, wseg, ipl, bx;
};

static BwGeom bw_geom(int N, int H, int W, int C, int K) {
    BwGeom g;
    g.gs = groups_per_slab(C);
    g.rt = 256 / (g.gs * K);
    if (g.rt < 1) g.rt = 1;
    g.slabs = (C / 8) / g.gs;
    g.wseg = W <= 32 ? W : 32;
    const int nseg = (W + g.wseg - 1) / g.wseg;
    const int64_t items = (int64_t)N * H * nseg;
    // aim at <= ~1024/slabs blocks along x (enough waves to hide L2 latency, small enough partial buffers)
    int64_t target = 1024 / g.slabs;
    if (target < 64) target = 64;
    int64_t ipl = ceil_div64(items, (int64_t)g.rt * target);
    if (ipl < 1) ipl = 1;
    g.ipl = (int)ipl;
    g.bx = (int)ceil_div64(items, (int64_t)g.rt * g.ipl);
    if (g.bx < 1) g.bx = 1;
    return g;
}

template <class T, int K>
int launch_fwd(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int dil,
               int pad_t, int pad_l, int flip, hipStream_t s) {
    const int gs = groups_per_slab(C);
    const int pt = 256 / gs;
    const int slabs = (C / 8) / gs;
    const int64_t tiles = (int64_t)N * H * ((W + TW - 1) / TW);
    int64_t bx = ceil_div64(tiles, pt);
    const int64_t cap = 256 * 8 / slabs > 1 ? 256 * 8 / slabs : 1;
    if (bx > cap) bx = cap;
    const size_t lds = (size_t)K * K * gs * 8 * sizeof(float);
    if (dil == 1)
        hipLaunchKernelGGL((dwconv_fwd_kernel<T, K, true>), dim3((unsigned)bx, slabs), dim3(256), lds, s, (const T*)x, w, bias,
                           (const T*)add, (T*)y, N, H, W, C, dil, pad_t, pad_l, flip, gs, pt);
    else
        hipLaunchKernelGGL((dwconv_fwd_kernel<T, K, false>), dim3((unsigned)bx, slabs), dim3(256), lds, s, (const T*)x, w, bias,
                           (const T*)add, (T*)y, N, H, W, C, dil, pad_t, pad_l, flip, gs, pt);
    return iseg_check_launch("iseg_dwconv2d");
}

template <class T, int K>
void launch_bw(const void* x, const void* dy, float* ws, int N, int H, int W, int C, int dil, int pad_t, int pad_l, const BwGeom& g,
               hipStream_t s) {
    const size_t lds = (size_t)(K * K + 1) * g.gs * 8 * sizeof(float);
    if (dil == 1)
        hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, K, true>), dim3(g.bx, g.slabs), dim3(256), lds, s, (const T*)x, (const T*)dy, ws,
                           N, H, W, C, dil, pad_t, pad_l, g.gs, g.rt, g.wseg, g.ipl);
    else
        hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, K, false>), dim3(g.bx, g.slabs), dim3(256), lds, s, (const T*)x, (const T*)dy,
                           ws, N, H, W, C, dil, pad_t, pad_l, g.gs, g.rt, g.wseg, g.ipl);
}

}  // namespace

extern "C" int iseg_dwconv2d_fwd(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W,
                                 int C, int K, int dil, int pad_t, int pad_l, int flip, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && w && y, "iseg_dwconv2d_fwd: null pointer");
    ISEG_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "iseg_dwconv2d_fwd: C=%d must be a multiple of 8", C);
    ISEG_REQUIRE(K == 3 || K == 5 || K == 7, "iseg_dwconv2d_fwd: kernel size %d unsupported (3,5,7)", K);
    ISEG_REQUIRE(dil >= 1, "iseg_dwconv2d_fwd: dilation must be >= 1");
    ISEG_REQUIRE((int64_t)N * H * W < (1ll << 31), "iseg_dwconv2d_fwd: more than 2^31 pixels");
#define DW_FWD(T)                                                                                          \
    (K == 7   ? launch_fwd<T, 7>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, stream)          \
     : K == 5 ? launch_fwd<T, 5>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, stream)          \
              : launch_fwd<T, 3>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, stream))
    return dtype == ISEG_BF16 ? DW_FWD(bf16_t) : DW_FWD(float);
#undef DW_FWD
}

extern "C" size_t iseg_dwconv2d_bwd_weight_workspace_bytes(int N, int H, int W, int C, int K) {
    const BwGeom g = bw_geom(N, H, W, C, K);
    return (size_t)g.bx * (K * K + 1) * C * sizeof(float);
}

extern "C" int iseg_dwconv2d_bwd_weight(const void* x, const void* dy, float* dw, float* db, int accumulate, int N, int H, int W,
                                        int C, int K, int dil, int pad_t, int pad_l, int dtype, void* ws, size_t ws_bytes,
                                        hipStream_t stream) {
    ISEG_REQUIRE(x && dy && dw, "iseg_dwconv2d_bwd_weight: null pointer");
    ISEG_REQUIRE(C % 8 == 0, "iseg_dwconv2d_bwd_weight: C=%d must be a multiple of 8", C);
    ISEG_REQUIRE(K == 3 || K == 5 || K == 7, "iseg_dwconv2d_bwd_weight: kernel size %d unsupported", K);
    ISEG_REQUIRE((int64_t)N * H * W < (1ll << 31), "iseg_dwconv2d_bwd_weight: more than 2^31 pixels");
    const BwGeom g = bw_geom(N, H, W, C, K);
    ISEG_REQUIRE(g.gs * K * g.rt <= 256, "iseg_dwconv2d_bwd_weight: slab does not fit a block");
    const size_t need = (size_t)g.bx * (K * K + 1) * C * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_dwconv2d_bwd_weight: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    if (dtype == ISEG_BF16) {
        if (K == 7) launch_bw<bf16_t, 7>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
        else if (K == 5) launch_bw<bf16_t, 5>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
        else launch_bw<bf16_t, 3>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
    } else {
        if (K == 7) launch_bw<float, 7>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
        else if (K == 5) launch_bw<float, 5>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
        else launch_bw<float, 3>(x, dy, (float*)ws, N, H, W, C, dil, pad_t, pad_l, g, stream);
    }
    const int n = (K * K + 1) * C;
    launch_reduce_rows((const float*)ws, g.bx, n, 0, 1, n, dw, db, K * K * C, 0, 1.f, accumulate, stream);
    return iseg_check_launch("iseg_dwconv2d_bwd_weight");
}
