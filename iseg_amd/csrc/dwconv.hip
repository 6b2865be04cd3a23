// Depthwise KxK convolution on NHWC tensors (keras.layers.DepthwiseConv2D, padding="same", stride 1,
// dilation d) -- the 7x7 depthwise of backbones/convnext.py:25,50 (dilated by build_dilated_convnext :245-266)
// and the 3x3 depthwise of layers/dcn_v3/dcn_v3.py.  VALU/L1-bound (49 FMA per output element), so:
//   * a lane owns 8 consecutive channels (one 16-B load per pixel) and TW output pixels along W, and slides a
//     register window over the input row so each loaded pixel feeds up to KW taps;
//   * weights of the block's channel slab sit in LDS as fp32 (the fp32 master kernel is read directly);
//   * backward-data is the same kernel with the taps flipped and the complementary padding, and can add the
//     residual branch's gradient on the way out (dx = dres + dwconv^T(dy));
//   * backward-weight gives each lane one kernel row (KW taps x 8 channels of accumulators) and a strip of
//     image rows, then reduces lanes -> block (LDS) -> grid (fixed-order partial sums, deterministic).
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
    const int64_t tiles_total = (int64_t)N * H * wtiles;
    const int c0 = slab_c0 + cg * 8;
    float bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) bv[u] = bias ? bias[c0 + u] : 0.f;

    for (int64_t tile = (int64_t)blockIdx.x * pt + ptile; tile < tiles_total; tile += (int64_t)gridDim.x * pt) {
        const int wt = (int)(tile % wtiles);
        const int64_t nh = tile / wtiles;
        const int h = (int)(nh % H);
        const int n = (int)(nh / H);
        const int w0 = wt * TW;
        float acc[TW][8];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[t][u] = bv[u];
        const T* xn = x + (int64_t)n * H * W * C + c0;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int ih = h + kh * dil - pad_t;
            if (ih < 0 || ih >= H) continue;
            const T* xr = xn + (int64_t)ih * W * C;
            if (DIL1) {
                float xin[TW + K - 1][8];
#pragma unroll
                for (int s = 0; s < TW + K - 1; ++s) {
                    const int iw = w0 - pad_l + s;
                    if (iw >= 0 && iw < W) load8<T>(xr + (int64_t)iw * C, xin[s]);
                    else {
#pragma unroll
                        for (int u = 0; u < 8; ++u) xin[s][u] = 0.f;
                    }
                }
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    float wv[8];
                    load8<float>(wl + (kh * K + kw) * sc + cg * 8, wv);
#pragma unroll
                    for (int t = 0; t < TW; ++t)
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc[t][u] = fmaf(xin[t + kw][u], wv[u], acc[t][u]);
                }
            } else {
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    float wv[8];
                    load8<float>(wl + (kh * K + kw) * sc + cg * 8, wv);
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
                        const int iw = w0 + t + kw * dil - pad_l;
                        if (iw >= 0 && iw < W) {
                            float xv[8];
                            load8<T>(xr + (int64_t)iw * C, xv);
#pragma unroll
                            for (int u = 0; u < 8; ++u) acc[t][u] = fmaf(xv[u], wv[u], acc[t][u]);
                        }
                    }
                }
            }
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
// thread = (channel group, kernel row, row-lane); loops over image rows, slides along W.
template <class T, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                float* __restrict__ partials, int N, int H, int W, int C,
                                                                int dil, int pad_t, int pad_l, int gs, int rt) {
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
        const int64_t rows_total = (int64_t)N * H;
        for (int64_t row = (int64_t)blockIdx.x * rt + rl; row < rows_total; row += (int64_t)gridDim.x * rt) {
            const int h = (int)(row % H);
            const int n = (int)(row / H);
            const int ih = h + kh * dil - pad_t;
            const bool row_ok = ih >= 0 && ih < H;
            if (!row_ok && kh != 0) continue;
            const T* dyr = dy + (((int64_t)n * H + h) * W) * C + c0;
            const T* xr = x + (((int64_t)n * H + (row_ok ? ih : 0)) * W) * C + c0;
            for (int ow = 0; ow < W; ++ow) {
                float d[8];
                load8<T>(dyr + (int64_t)ow * C, d);
                if (kh == 0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) accb[u] += d[u];
                }
                if (row_ok) {
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const int iw = ow + j * dil - pad_l;
                        if (iw >= 0 && iw < W) {
                            float xv[8];
                            load8<T>(xr + (int64_t)iw * C, xv);
#pragma unroll
                            for (int u = 0; u < 8; ++u) acc[j][u] = fmaf(xv[u], d[u], acc[j][u]);
                        }
                    }
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

__global__ void dw_reduce_partials_kernel(const float* __restrict__ partials, int P, int taps, int C, float* __restrict__ dw,
                                          float* __restrict__ db, int accumulate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = (taps + 1) * C;
    if (j >= n) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += partials[(int64_t)p * n + j];
    float* dst = j < taps * C ? dw + j : (db ? db + (j - taps * C) : nullptr);
    if (!dst) return;
    if (accumulate) s += *dst;
    *dst = s;
}

static int bw_rows_blocks(int N, int H, int rt) {
    int64_t b = ceil_div64((int64_t)N * H, (int64_t)rt * 4);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
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

}  // namespace

extern "C" int iseg_dwconv2d_fwd(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W,
                                 int C, int K, int dil, int pad_t, int pad_l, int flip, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && w && y, "iseg_dwconv2d_fwd: null pointer");
    ISEG_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "iseg_dwconv2d_fwd: C=%d must be a multiple of 8", C);
    ISEG_REQUIRE(K == 3 || K == 5 || K == 7, "iseg_dwconv2d_fwd: kernel size %d unsupported (3,5,7)", K);
    ISEG_REQUIRE(dil >= 1, "iseg_dwconv2d_fwd: dilation must be >= 1");
#define DW_FWD(T)                                                                                          \
    (K == 7   ? launch_fwd<T, 7>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, stream)          \
     : K == 5 ? launch_fwd<T, 5>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, stream)          \
              : launch_fwd<T, 3>(x, w, bias, add, y, N, H, W, C, dil, pad_t, pad_l, flip, stream))
    return dtype == ISEG_BF16 ? DW_FWD(bf16_t) : DW_FWD(float);
#undef DW_FWD
}

extern "C" size_t iseg_dwconv2d_bwd_weight_workspace_bytes(int N, int H, int C, int K) {
    const int gs = groups_per_slab(C);
    int rt = 256 / (gs * K);
    if (rt < 1) rt = 1;
    return (size_t)bw_rows_blocks(N, H, rt) * (K * K + 1) * C * sizeof(float);
}

extern "C" int iseg_dwconv2d_bwd_weight(const void* x, const void* dy, float* dw, float* db, int accumulate, int N, int H, int W,
                                        int C, int K, int dil, int pad_t, int pad_l, int dtype, void* ws, size_t ws_bytes,
                                        hipStream_t stream) {
    ISEG_REQUIRE(x && dy && dw, "iseg_dwconv2d_bwd_weight: null pointer");
    ISEG_REQUIRE(C % 8 == 0, "iseg_dwconv2d_bwd_weight: C=%d must be a multiple of 8", C);
    ISEG_REQUIRE(K == 3 || K == 5 || K == 7, "iseg_dwconv2d_bwd_weight: kernel size %d unsupported", K);
    const int gs = groups_per_slab(C);
    int rt = 256 / (gs * K);
    if (rt < 1) rt = 1;
    ISEG_REQUIRE(gs * K * rt <= 256, "iseg_dwconv2d_bwd_weight: slab does not fit a block");
    const int slabs = (C / 8) / gs;
    const int bx = bw_rows_blocks(N, H, rt);
    const size_t need = (size_t)bx * (K * K + 1) * C * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_dwconv2d_bwd_weight: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const size_t lds = (size_t)(K * K + 1) * gs * 8 * sizeof(float);
#define DW_BW(T, KK)                                                                                                       \
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, KK>), dim3(bx, slabs), dim3(256), lds, stream, (const T*)x, (const T*)dy, \
                       (float*)ws, N, H, W, C, dil, pad_t, pad_l, gs, rt)
    if (dtype == ISEG_BF16) {
        if (K == 7) DW_BW(bf16_t, 7);
        else if (K == 5) DW_BW(bf16_t, 5);
        else DW_BW(bf16_t, 3);
    } else {
        if (K == 7) DW_BW(float, 7);
        else if (K == 5) DW_BW(float, 5);
        else DW_BW(float, 3);
    }
#undef DW_BW
    const int n = (K * K + 1) * C;
    hipLaunchKernelGGL(dw_reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float*)ws, bx, K * K, C, dw,
                       db, accumulate);
    return iseg_check_launch("iseg_dwconv2d_bwd_weight");
}
