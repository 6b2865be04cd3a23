// Strided depthwise convolution (gfx950): keras.layers.DepthwiseConv2D(strides = s, padding = "same", dilation_rate = d) and both gradients,
// computed at the strided output positions only.
//
// Replaces the stride-2 depthwise layers of the reference's separable / inverted-residual families (backbones/mobilenetv2.py:60-78 the
// expanded 3x3 / s2 of an inverted residual block, backbones/xception.py / layers/model_builder.py:200-250 SepConvBnReLU with a stride),
// which round 2 served as the stride-1 result sampled by a row gather (s^2 times the arithmetic).  HBM-bound streaming kernels: a lane owns
// eight consecutive channels of one pixel (16-byte bf16 / 32-byte fp32 accesses, NHWC), the K x K x 8 weights come from the scalar-cached
// [tap][C] fp32 table.
//
//   forward      y[n, oh, ow, c]  = b[c] + sum_{i,j} x[n, oh*s + i*d - pt, ow*s + j*d - pl, c] * w[i, j, c]
//   data grad    dx[n, h, w, c]   = sum_{i,j : (h + pt - i*d) % s == 0, (w + pl - j*d) % s == 0} dy[n, (h + pt - i*d)/s, (w + pl - j*d)/s, c] * w[i, j, c]
//   weight grad  dw[i, j, c]      = sum_{n,oh,ow} x[n, oh*s + i*d - pt, ow*s + j*d - pl, c] * dy[n, oh, ow, c];   db[c] = sum dy
//                (per-workgroup partial sums over a fixed pixel partition, lanes of one channel combined in a fixed order through LDS, then the
//                fixed-order row reduction shared with the stride-1 kernels: deterministic)
#include <type_traits>

#include "common.h"

namespace {

struct SGeom {
    int N, H, W, C, Ho, Wo, s, d, pt, pl;
};

template <class T, int K>
__global__ __launch_bounds__(256) void dws_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      T* __restrict__ y, SGeom g, int64_t total) {
    const int cg = g.C / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cg) * 8;
        int64_t pix = i / cg;
        const int ow = (int)(pix % g.Wo);
        pix /= g.Wo;
        const int oh = (int)(pix % g.Ho), n = (int)(pix / g.Ho);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = bias ? bias[c + u] : 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int ih = oh * g.s + ky * g.d - g.pt;
            if (ih < 0 || ih >= g.H) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int iw = ow * g.s + kx * g.d - g.pl;
                if (iw < 0 || iw >= g.W) continue;
                float xv[8], wv[8];
                load8<T>(x + (((int64_t)n * g.H + ih) * g.W + iw) * g.C + c, xv);
                load8<float>(w + (ky * K + kx) * g.C + c, wv);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = fmaf(xv[u], wv[u], acc[u]);
            }
        }
        store8<T>(y + i * 8, acc);
    }
}

template <class T, int K>
__global__ __launch_bounds__(256) void dws_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, SGeom g,
                                                           int64_t total) {
    const int cg = g.C / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cg) * 8;
        int64_t pix = i / cg;
        const int iw = (int)(pix % g.W);
        pix /= g.W;
        const int ih = (int)(pix % g.H), n = (int)(pix / g.H);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int th = ih + g.pt - ky * g.d;
            if (th < 0 || th % g.s != 0 || th / g.s >= g.Ho) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int tw = iw + g.pl - kx * g.d;
                if (tw < 0 || tw % g.s != 0 || tw / g.s >= g.Wo) continue;
                float dv[8], wv[8];
                load8<T>(dy + (((int64_t)n * g.Ho + th / g.s) * g.Wo + tw / g.s) * g.C + c, dv);
                load8<float>(w + (ky * K + kx) * g.C + c, wv);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = fmaf(dv[u], wv[u], acc[u]);
            }
        }
        store8<T>(dx + i * 8, acc);
    }
}

// workgroup = 32 channels (lane & 31) x 8 pixel lanes; blockIdx.y = channel slab, blockIdx.x = pixel partition.  Partial layout per
// partition: [K*K + 1][C] (taps, then the bias row), as the stride-1 kernels write it.
template <class T, int K>
__global__ __launch_bounds__(256) void dws_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                             SGeom g, int64_t pixels) {
    __shared__ float red[8][K * K + 1][32];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.y * 32 + cl;
    const bool live = c < g.C;
    float acc[K * K + 1];
#pragma unroll
    for (int t = 0; t <= K * K; ++t) acc[t] = 0.f;
    if (live) {
        for (int64_t pix = (int64_t)blockIdx.x * 8 + pl; pix < pixels; pix += (int64_t)gridDim.x * 8) {
            const int ow = (int)(pix % g.Wo);
            const int64_t r = pix / g.Wo;
            const int oh = (int)(r % g.Ho), n = (int)(r / g.Ho);
            const float d = (float)dy[pix * g.C + c];
            acc[K * K] += d;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int ih = oh * g.s + ky * g.d - g.pt;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int iw = ow * g.s + kx * g.d - g.pl;
                    const bool ok = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
                    const float xv = ok ? (float)x[(((int64_t)n * g.H + ih) * g.W + iw) * g.C + c] : 0.f;
                    acc[ky * K + kx] = fmaf(xv, d, acc[ky * K + kx]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t <= K * K; ++t) red[pl][t][cl] = acc[t];
    __syncthreads();
    for (int e = threadIdx.x; e < (K * K + 1) * 32; e += 256) {
        const int t = e / 32, l = e % 32;
        if (blockIdx.y * 32 + l >= g.C) continue;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += red[q][t][l];      // fixed order
        partial[((int64_t)blockIdx.x * (K * K + 1) + t) * g.C + blockIdx.y * 32 + l] = s;
    }
}

int bw_partitions(int64_t pixels, int C) {
    const int slabs = (C + 31) / 32;
    int64_t p = 2048 / slabs;      // ~8 workgroups per CU over the whole launch
    if (p < 16) p = 16;
    const int64_t most = (pixels + 7) / 8;
    if (p > most) p = most;
    return (int)(p < 1 ? 1 : p);
}

bool geom_ok(int N, int H, int W, int C, int K, int s, int d, int Ho, int Wo) {
    return N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (K == 3 || K == 5 || K == 7) && s >= 1 && d >= 1 && Ho > 0 && Wo > 0 &&
           (int64_t)N * H * W * C < (1ll << 31);
}

template <class F> int by_dtype_k(int dtype, int K, F&& f) {
    // (generic lambdas over <T, K> keep the six instantiations in one place)
    if (dtype == ISEG_BF16) {
        if (K == 3) return f(bf16_t{}, std::integral_constant<int, 3>{});
        if (K == 5) return f(bf16_t{}, std::integral_constant<int, 5>{});
        return f(bf16_t{}, std::integral_constant<int, 7>{});
    }
    if (K == 3) return f(float{}, std::integral_constant<int, 3>{});
    if (K == 5) return f(float{}, std::integral_constant<int, 5>{});
    return f(float{}, std::integral_constant<int, 7>{});
}

}  // namespace

extern "C" int iseg_dwconv2d_strided_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C, int K, int stride,
                                         int dil, int pad_t, int pad_l, int Ho, int Wo, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && w && y, "iseg_dwconv2d_strided_fwd: null pointer");
    ISEG_REQUIRE(geom_ok(N, H, W, C, K, stride, dil, Ho, Wo), "iseg_dwconv2d_strided_fwd: needs C %% 8 == 0 and K in (3, 5, 7) (C %d, K %d)", C, K);
    ISEG_REQUIRE(dtype == ISEG_BF16 || dtype == ISEG_F32, "iseg_dwconv2d_strided_fwd: dtype %d", dtype);
    const SGeom g{N, H, W, C, Ho, Wo, stride, dil, pad_t, pad_l};
    const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    return by_dtype_k(dtype, K, [&](auto t, auto k) {
        using T = decltype(t);
        hipLaunchKernelGGL((dws_fwd_kernel<T, decltype(k)::value>), dim3(grid), dim3(256), 0, stream, (const T*)x, w, bias, (T*)y, g, total);
        return iseg_check_launch("iseg_dwconv2d_strided_fwd");
    });
}

extern "C" int iseg_dwconv2d_strided_bwd_data(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int K, int stride, int dil,
                                              int pad_t, int pad_l, int Ho, int Wo, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(dy && w && dx, "iseg_dwconv2d_strided_bwd_data: null pointer");
    ISEG_REQUIRE(geom_ok(N, H, W, C, K, stride, dil, Ho, Wo), "iseg_dwconv2d_strided_bwd_data: needs C %% 8 == 0 and K in (3, 5, 7) (C %d, K %d)", C, K);
    ISEG_REQUIRE(dtype == ISEG_BF16 || dtype == ISEG_F32, "iseg_dwconv2d_strided_bwd_data: dtype %d", dtype);
    const SGeom g{N, H, W, C, Ho, Wo, stride, dil, pad_t, pad_l};
    const int64_t total = (int64_t)N * H * W * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    return by_dtype_k(dtype, K, [&](auto t, auto k) {
        using T = decltype(t);
        hipLaunchKernelGGL((dws_bwd_data_kernel<T, decltype(k)::value>), dim3(grid), dim3(256), 0, stream, (const T*)dy, w, (T*)dx, g, total);
        return iseg_check_launch("iseg_dwconv2d_strided_bwd_data");
    });
}

extern "C" size_t iseg_dwconv2d_strided_bwd_weight_workspace_bytes(int N, int Ho, int Wo, int C, int K) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || K <= 0) return 0;
    return (size_t)bw_partitions((int64_t)N * Ho * Wo, C) * (K * K + 1) * C * sizeof(float);
}

extern "C" int iseg_dwconv2d_strided_bwd_weight(const void* x, const void* dy, float* dw, float* db, int accumulate, int N, int H, int W, int C,
                                                int K, int stride, int dil, int pad_t, int pad_l, int Ho, int Wo, int dtype, void* ws,
                                                size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(x && dy && dw, "iseg_dwconv2d_strided_bwd_weight: null pointer");
    ISEG_REQUIRE(geom_ok(N, H, W, C, K, stride, dil, Ho, Wo), "iseg_dwconv2d_strided_bwd_weight: needs C %% 8 == 0 and K in (3, 5, 7) (C %d, K %d)", C, K);
    ISEG_REQUIRE(dtype == ISEG_BF16 || dtype == ISEG_F32, "iseg_dwconv2d_strided_bwd_weight: dtype %d", dtype);
    const int64_t pixels = (int64_t)N * Ho * Wo;
    const int parts = bw_partitions(pixels, C);
    const size_t need = (size_t)parts * (K * K + 1) * C * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_dwconv2d_strided_bwd_weight: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const SGeom g{N, H, W, C, Ho, Wo, stride, dil, pad_t, pad_l};
    const int rc = by_dtype_k(dtype, K, [&](auto t, auto k) {
        using T = decltype(t);
        hipLaunchKernelGGL((dws_bwd_weight_kernel<T, decltype(k)::value>), dim3(parts, (C + 31) / 32), dim3(256), 0, stream, (const T*)x, (const T*)dy,
                           (float*)ws, g, pixels);
        return iseg_check_launch("iseg_dwconv2d_strided_bwd_weight");
    });
    if (rc != ISEG_OK) return rc;
    const int n = (K * K + 1) * C;
    launch_reduce_rows((const float*)ws, parts, n, 0, 1, n, dw, db, (int64_t)K * K * C, 0, 1.f, accumulate, stream);
    return iseg_check_launch("iseg_dwconv2d_strided_bwd_weight");
}
