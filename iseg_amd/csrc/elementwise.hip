// HBM-bound data-movement and elementwise kernels: dtype casts, im2col / col2im for the few real spatial
// convolutions (stem 4x4/s4, downsample 2x2/s2, ASPP 3x3 dilated: backbones/convnext.py:72-75, layers/aspp.py:41-52),
// column sums (bias / broadcast gradients), global average pooling + broadcast (layers/model_builder.py:260-273
// ImageLevelBlock), drop-path / dropout masks (utils/drops.py:8-22, keras.layers.Dropout), residual adds.
// All of them are one pass over the data with 16-B lanes and grid-stride loops capped at 8 blocks per CU.
#include "common.h"
#include "iseg_hip.h"

namespace {

static inline unsigned cap_blocks(int64_t work_items, int per_block = 256) {
    int64_t b = ceil_div64(work_items, per_block);
    if (b > 256 * 8) b = 256 * 8;
    if (b < 1) b = 1;
    return (unsigned)b;
}

template <class TI, class TO>
__global__ void cast_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int64_t n) {
    const int64_t nv = n / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        load8<TI>(src + i * 8, v);
        store8<TO>(dst + i * 8, v);
    }
    for (int64_t i = nv * 8 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = from_f32<TO>(to_f32(src[i]));
}

// dst[k][n] = src[k][n] * colscale[n]  (f32 master -> compute dtype), used for the layer-scale folded weights
template <class TO>
__global__ void scale_cols_cast_kernel(const float* __restrict__ src, const float* __restrict__ colscale, TO* __restrict__ dst,
                                       int64_t rows, int cols) {
    const int64_t total = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = from_f32<TO>(src[i] * colscale[i % cols]);
}

// ---------------------------------------------------------------------------------------------
// im2col: col[m][ (i*KW + j)*C + c ] = x[n, oh*sh + i*dh - pt, ow*sw + j*dw - pl, c]   (0 outside)
// ---------------------------------------------------------------------------------------------
template <class TI, class TO, int V>
__global__ void im2col_kernel(const TI* __restrict__ x, TO* __restrict__ col, int N, int H, int W, int C, int KH, int KW, int sh,
                              int sw, int dh, int dw, int pt, int pl, int Ho, int Wo, int64_t ldc) {
    const int cv = C / V;
    const int64_t kk = (int64_t)KH * KW * cv;
    const int64_t M = (int64_t)N * Ho * Wo;
    const int64_t total = M * kk;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) * V;
        int64_t r = idx / cv;
        const int j = (int)(r % KW);
        r /= KW;
        const int i = (int)(r % KH);
        const int64_t m = r / KH;
        const int ow = (int)(m % Wo);
        const int oh = (int)((m / Wo) % Ho);
        const int n = (int)(m / ((int64_t)Wo * Ho));
        const int ih = oh * sh + i * dh - pt, iw = ow * sw + j * dw - pl;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = 0.f;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
            const TI* p = x + (((int64_t)n * H + ih) * W + iw) * C + c;
            if (V == 8) load8<TI>(p, v);
            else v[0] = to_f32(p[0]);
        }
        TO* q = col + m * ldc + ((int64_t)i * KW + j) * C + c;
        if (V == 8) store8<TO>(q, v);
        else q[0] = from_f32<TO>(v[0]);
    }
}

// zero the padding columns [K, ldc) of every row
template <class TO>
__global__ void zero_pad_cols_kernel(TO* __restrict__ col, int64_t M, int64_t K, int64_t ldc) {
    const int64_t padw = ldc - K;
    const int64_t total = M * padw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        col[(i / padw) * ldc + K + (i % padw)] = from_f32<TO>(0.f);
}

// col2im (gather form, deterministic): dx[n,ih,iw,c] = sum over taps (i,j) and outputs (oh,ow) that read it
template <class T, int V>
__global__ void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int N, int H, int W, int C, int KH, int KW, int sh,
                              int sw, int dh, int dw, int pt, int pl, int Ho, int Wo, int64_t ldc) {
    const int cv = C / V;
    const int64_t total = (int64_t)N * H * W * cv;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) * V;
        int64_t r = idx / cv;
        const int iw = (int)(r % W);
        r /= W;
        const int ih = (int)(r % H);
        const int n = (int)(r / H);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
        for (int i = 0; i < KH; ++i) {
            const int th = ih + pt - i * dh;
            if (th < 0 || th % sh != 0) continue;
            const int oh = th / sh;
            if (oh >= Ho) continue;
            for (int j = 0; j < KW; ++j) {
                const int tw = iw + pl - j * dw;
                if (tw < 0 || tw % sw != 0) continue;
                const int ow = tw / sw;
                if (ow >= Wo) continue;
                const T* p = dcol + (((int64_t)n * Ho + oh) * Wo + ow) * ldc + ((int64_t)i * KW + j) * C + c;
                if (V == 8) {
                    float v[8];
                    load8<T>(p, v);
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[u] += v[u];
                } else {
                    acc[0] += to_f32(p[0]);
                }
            }
        }
        T* q = dx + idx / cv * C + c;
        if (V == 8) store8<T>(q, acc);
        else q[0] = from_f32<T>(acc[0]);
    }
}

// ---------------------------------------------------------------------------------------------
// batched column sums: out[b][c] = scale * sum_r x[b][r][c]   (x row stride ldx, batch stride bsx)
// ---------------------------------------------------------------------------------------------
template <class T, int V>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int64_t ldx, int64_t bsx,
                                                             float* __restrict__ partials, int64_t rows, int C) {
    extern __shared__ __attribute__((aligned(16))) float lds_s[];  // [rows per iteration][C]: one slab row per row lane, summed in row order
    const int b = blockIdx.y;
    const int nch = C / V;
    const int tpc = nch < 256 ? nch : 256;
    const int rpi = 256 / tpc;
    const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
    const T* xb = x + (int64_t)b * bsx;
    if (tr < rpi) {
        for (int c = tc; c < nch; c += tpc) {
            float s[8] = {};
            int64_t r = (int64_t)blockIdx.x * rpi + tr;
            const int64_t rstep = (int64_t)gridDim.x * rpi;
            if (V == 8) {
                // eight rows per trip, their loads issued together: one dependent load per trip left every lane waiting a full
                // L2 round trip per row (14.7 us for a 12.6 MB tensor)
                constexpr int UB = 8;
                for (; r + (UB - 1) * rstep < rows; r += UB * rstep) {
                    float v[UB][8];
#pragma unroll
                    for (int q = 0; q < UB; ++q) load8<T>(xb + (r + q * rstep) * ldx + c * 8, v[q]);
#pragma unroll
                    for (int q = 0; q < UB; ++q)
#pragma unroll
                        for (int u = 0; u < 8; ++u) s[u] += v[q][u];
                }
            }
            for (; r < rows; r += rstep) {
                if (V == 8) {
                    float v[8];
                    load8<T>(xb + r * ldx + c * 8, v);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s[u] += v[u];
                } else {
                    s[0] += to_f32(xb[r * ldx + c]);
                }
            }
            // (no float atomics: every cell of slab rows 0..rpi-1 is written exactly once, the sum below runs in row order -- bit-reproducible)
#pragma unroll
            for (int u = 0; u < V; ++u) lds_s[(size_t)tr * C + c * V + u] = s[u];
        }
    }
    __syncthreads();
    float* out = partials + ((int64_t)b * gridDim.x + blockIdx.x) * C;
    for (int i = threadIdx.x; i < C; i += 256) {
        float a = lds_s[i];
        for (int t = 1; t < rpi; ++t) a += lds_s[(size_t)t * C + i];
        out[i] = a;
    }
}

static size_t colsum_slab_bytes(int C, int V) {
    const int nch = C / V;
    const int tpc = nch < 256 ? nch : 256;
    return (size_t)(256 / tpc) * C * sizeof(float);
}

static int colsum_blocks(int64_t rows, int C, int V) {
    const int nch = C / V;
    const int tpc = nch < 256 ? nch : 256;
    const int rpi = 256 / tpc;
    int64_t b = ceil_div64(rows, (int64_t)rpi * 8);
    if (b > 256) b = 256;
    if (b < 1) b = 1;
    return (int)b;
}

// y[b][r][c] (+)= scale * v[b][c]     (broadcast a per-sample vector over R rows; y row stride ldy, batch stride bsy)
template <class T, class TV>
__global__ void broadcast_rows_kernel(const TV* __restrict__ v, T* __restrict__ y, int64_t ldy, int64_t bsy, int B, int64_t R,
                                      int C, float scale, int accumulate) {
    const int cv = C / 8;
    const int64_t total = (int64_t)B * R * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 8;
        const int64_t r = (i / cv) % R;
        const int b = (int)(i / (cv * R));
        float a[8];
        load8<TV>(v + (int64_t)b * C + c, a);
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] *= scale;
        T* p = y + (int64_t)b * bsy + r * ldy + c;
        if (accumulate) {
            float o[8];
            load8<T>(p, o);
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += o[u];
        }
        store8<T>(p, a);
    }
}

// y = alpha*a + beta*b   (b may be null)
template <class T>
__global__ void axpby_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, float alpha, float beta,
                             int64_t n) {
    const int64_t nv = n / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float va[8], vb[8];
        load8<T>(a + i * 8, va);
        if (b) {
            load8<T>(b + i * 8, vb);
#pragma unroll
            for (int u = 0; u < 8; ++u) va[u] = alpha * va[u] + beta * vb[u];
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) va[u] = alpha * va[u];
        }
        store8<T>(y + i * 8, va);
    }
    for (int64_t i = nv * 8 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = from_f32<T>(alpha * to_f32(a[i]) + (b ? beta * to_f32(b[i]) : 0.f));
}

// dst0[i] += src[i], dst1[i] += src[n + i]  (either destination may be null): the two parameter gradients a normalisation layer's packed
// backward sums carry (dbeta | dgamma), in one launch
__global__ __launch_bounds__(256) void accumulate_pair_kernel(const float* __restrict__ src, int n, float* __restrict__ dst0, float* __restrict__ dst1) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += gridDim.x * 256) {
        float* d = i < n ? dst0 : dst1;
        if (d) d[i < n ? i : i - n] += src[i];
    }
}

template <class T>
__global__ void scale_dev_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y, int64_t n) {
    const float f = s[0];
    const int64_t nv = n / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] *= f;
        store8<T>(y + i * 8, v);
    }
    for (int64_t i = nv * 8 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = from_f32<T>(to_f32(x[i]) * f);
}

// y[m][:] = x[m][:] * s[m / rows_per_group]
template <class T>
__global__ void rowscale_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y, int64_t rows, int C,
                                int64_t rows_per_group) {
    const int cv = C / 8;
    const int64_t total = rows * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / cv;
        const float f = s[m / rows_per_group];
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] *= f;
        store8<T>(y + i * 8, v);
    }
}

// counter-based uniform in [0,1): splitmix64 of (seed, index)
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// y = x * (u >= rate) / (1-rate); the mask is a pure function of (seed, element index): backward re-derives it
template <class T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float rate, uint64_t seed, const uint64_t* __restrict__ seed_offset) {
    if (seed_offset) seed += *seed_offset;      // (a captured step: the launch arguments are frozen, the draw counter moves on in device memory)
    const float inv_keep = 1.0f / (1.0f - rate);
    const int64_t nv = n / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = uniform01(seed, (uint64_t)(i * 8 + u)) >= rate ? v[u] * inv_keep : 0.f;
        store8<T>(y + i * 8, v);
    }
    for (int64_t i = nv * 8 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = from_f32<T>(uniform01(seed, (uint64_t)i) >= rate ? to_f32(x[i]) * inv_keep : 0.f);
}

// per-sample drop-path factors: s[n] = floor(keep + u_n) / keep   (utils/drops.py:14-20)
__global__ void drop_path_mask_kernel(float* __restrict__ s, int n, float keep, uint64_t seed, const uint64_t* __restrict__ seed_offset) {
    if (seed_offset) seed += *seed_offset;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) s[i] = floorf(keep + uniform01(seed, (uint64_t)i)) / keep;
}

// the masks of a whole training step in one launch: row p (one per drop_path call site, in call order) uses its own keep probability
// and its own stream of the counter-based generator
__global__ void drop_path_masks_kernel(float* __restrict__ s, const float* __restrict__ keep, int P, int n, uint64_t seed, const uint64_t* __restrict__ seed_offset) {
    if (seed_offset) seed += *seed_offset;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * n) return;
    const int p = i / n;
    const float k = keep[p];
    s[i] = floorf(k + uniform01(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(p + 1), (uint64_t)(i - p * n))) / k;
}

__global__ void fill_kernel(float* __restrict__ p, float v, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void rsqrt_eps_kernel(const float* __restrict__ var, float eps, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rsqrtf(var[i] + eps);
}

// standalone activations (keras.activations.relu / gelu / sigmoid / swish) for layers whose activation cannot ride a GEMM epilogue.
// FAST (bf16 storage): the transcendental-free GELU forms of common.h; fp32 storage: libm erf.  sigmoid / swish (tf.nn.sigmoid, tf.nn.silu:
// layers/nasfpn.py:304-311, backbones/eva/swiglu.py:13) are evaluated with expf in both.
template <bool FAST> __device__ __forceinline__ float act_value(float v, int act) {
    switch (act) {
        case ISEG_ACT_RELU: return fmaxf(v, 0.f);
        case ISEG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case ISEG_ACT_SWISH: return v / (1.f + expf(-v));
        default: return FAST ? gelu_poly(v) : gelu_erf(v);
    }
}
// d act / d pre-activation at the pre-activation a
template <bool FAST> __device__ __forceinline__ float act_deriv(float a, int act) {
    switch (act) {
        case ISEG_ACT_RELU: return a > 0.f ? 1.f : 0.f;
        case ISEG_ACT_SIGMOID: {
            const float s = 1.f / (1.f + expf(-a));
            return s * (1.f - s);
        }
        case ISEG_ACT_SWISH: {
            const float s = 1.f / (1.f + expf(-a));
            return s * (1.f + a * (1.f - s));
        }
        default: return FAST ? gelu_poly_grad(a) : gelu_erf_grad(a);
    }
}

template <class T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int act) {
    const int64_t nv = n / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = act_value<sizeof(T) == 2>(v[u], act);
        store8<T>(y + i * 8, v);
    }
    for (int64_t i = nv * 8 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = from_f32<T>(act_value<false>(to_f32(x[i]), act));
}

// dx = dy * act'(aux): aux = the pre-activation (for relu the post-activation serves as well)
template <class T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ aux, T* __restrict__ dx, int64_t n, int act) {
    const int64_t nv = n / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float d[8], a[8];
        load8<T>(dy + i * 8, d);
        load8<T>(aux + i * 8, a);
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = act == ISEG_ACT_RELU ? (a[u] > 0.f ? d[u] : 0.f) : d[u] * act_deriv<sizeof(T) == 2>(a[u], act);
        store8<T>(dx + i * 8, d);
    }
    for (int64_t i = nv * 8 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = to_f32(dy[i]), a = to_f32(aux[i]);
        dx[i] = from_f32<T>(act == ISEG_ACT_RELU ? (a > 0.f ? d : 0.f) : d * act_deriv<false>(a, act));
    }
}

// dst[r][0:cols] = src[r][0:cols] with independent row strides (tf.concat along channels writes slices in place)
template <class T>
__global__ void copy2d_kernel(const T* __restrict__ src, int64_t lds_, T* __restrict__ dst, int64_t ldd, int64_t rows, int cols) {
    const int cv = cols / 8;
    const int64_t total = rows * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cv;
        const int c = (int)(i % cv) * 8;
        float v[8];
        load8<T>(src + r * lds_ + c, v);
        store8<T>(dst + r * ldd + c, v);
    }
}

// layer-scale bookkeeping for one ConvNeXt block (backbones/convnext.py:56-57), from Z = g^T @ dout (unscaled):
//   dW2[k][n] = Z[k][n]*gamma[n] ; dgamma[n] = sum_k W2[k][n]*Z[k][n] + b2[n]*S[n] ; db2[n] = gamma[n]*S[n], S = colsum(dout)
// stage 1: elementwise dW2, db2, and per-block partial column dots (block = 64 columns x 4 k-lanes, strip of k rows); stage 2 = the common
// row reduction of the partials into dgamma (reduce_rows, or the deferred queue when dgamma lies in the gradient buffer)
__global__ __launch_bounds__(256) void layerscale_stage1_kernel(const float* __restrict__ Z, const float* __restrict__ W2,
                                                                const float* __restrict__ gamma, const float* __restrict__ b2,
                                                                const float* __restrict__ S, float* __restrict__ dW2,
                                                                float* __restrict__ db2, float* __restrict__ partials, int Kdim, int Ndim,
                                                                int accumulate) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (n < Ndim) {
        const float g = gamma[n];
        for (int k = blockIdx.y * 4 + ty; k < Kdim; k += gridDim.y * 4) {
            const int64_t o = (int64_t)k * Ndim + n;
            const float z = Z[o];
            s += W2[o] * z;
            const float v = z * g;
            if (accumulate) dW2[o] += v;
            else dW2[o] = v;
        }
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < Ndim) {
        float part = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
        if (blockIdx.y == 0) {      // the S terms ride the first partial row, so the second stage is a plain row reduction (deferrable)
            part += b2[n] * S[n];
            const float dbv = gamma[n] * S[n];
            if (accumulate) db2[n] += dbv;
            else db2[n] = dbv;
        }
        partials[(int64_t)blockIdx.y * Ndim + n] = part;
    }
}

// The same bookkeeping straight from the split-K slabs of the weight-gradient product Z = g^T dout (iseg_gemm with a deferred reduction:
// slabs [nslabs][K + 1][N] fp32, row K = the virtual ones-row = column sums of dout): Z and S are summed in slab order while they are loaded,
// so neither the slab-sum launch nor the Z tensor exists.  A lane owns four consecutive columns (16-byte loads), blocks are 16 columns-quads x
// 16 row lanes over a strip of rows; partial column dots per block row as in the kernel above.
__device__ __forceinline__ void layerscale_slabs_body(const float* __restrict__ slabs, int nslabs, int64_t slab_stride,
                                                      const float* __restrict__ W2, const float* __restrict__ gamma,
                                                      const float* __restrict__ b2, float* __restrict__ dW2, float* __restrict__ db2,
                                                      float* __restrict__ partials, int Kdim, int Ndim, int accumulate, int bx, int by, int gy,
                                                      bool srow = false) {
    __shared__ float4 red[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n = (bx * 16 + tx) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < Ndim) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + n);
        for (int k = by * 16 + ty; k < Kdim; k += gy * 16) {
            const int64_t o = (int64_t)k * Ndim + n;
            float4 z = *reinterpret_cast<const float4*>(slabs + o);
            for (int q = 1; q < nslabs; ++q) {      // slab order: the sum the reduction kernel would have formed
                const float4 t = *reinterpret_cast<const float4*>(slabs + q * slab_stride + o);
                z.x += t.x, z.y += t.y, z.z += t.z, z.w += t.w;
            }
            const float4 w = *reinterpret_cast<const float4*>(W2 + o);
            s.x = fmaf(w.x, z.x, s.x), s.y = fmaf(w.y, z.y, s.y), s.z = fmaf(w.z, z.z, s.z), s.w = fmaf(w.w, z.w, s.w);
            float4 v = make_float4(z.x * g.x, z.y * g.y, z.z * g.z, z.w * g.w);
            if (accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(dW2 + o);
                v.x += old.x, v.y += old.y, v.z += old.z, v.w += old.w;
            }
            *reinterpret_cast<float4*>(dW2 + o) = v;
        }
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < Ndim) {
        float4 part = red[0][tx];
#pragma unroll
        for (int i = 1; i < 16; ++i) part.x += red[i][tx].x, part.y += red[i][tx].y, part.z += red[i][tx].z, part.w += red[i][tx].w;
        if (by == 0 && !srow) {      // the S terms ride the first partial row (S = the ones-row of the slabs, summed in slab order)
            const int64_t o = (int64_t)Kdim * Ndim + n;
            float4 S = *reinterpret_cast<const float4*>(slabs + o);
            for (int q = 1; q < nslabs; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(slabs + q * slab_stride + o);
                S.x += t.x, S.y += t.y, S.z += t.z, S.w += t.w;
            }
            const float4 g = *reinterpret_cast<const float4*>(gamma + n), b = *reinterpret_cast<const float4*>(b2 + n);
            part.x = fmaf(b.x, S.x, part.x), part.y = fmaf(b.y, S.y, part.y), part.z = fmaf(b.z, S.z, part.z), part.w = fmaf(b.w, S.w, part.w);
            float4 dbv = make_float4(g.x * S.x, g.y * S.y, g.z * S.z, g.w * S.w);
            if (accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(db2 + n);
                dbv.x += old.x, dbv.y += old.y, dbv.z += old.z, dbv.w += old.w;
            }
            *reinterpret_cast<float4*>(db2 + n) = dbv;
        }
        if (srow) {      // partial rows are [dgamma part | db2 part]: the S terms come from the SRowJob workgroups' rows
            *reinterpret_cast<float4*>(partials + (int64_t)by * 2 * Ndim + n) = part;
            *reinterpret_cast<float4*>(partials + (int64_t)by * 2 * Ndim + Ndim + n) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            *reinterpret_cast<float4*>(partials + (int64_t)by * Ndim + n) = part;
        }
    }
}

__global__ __launch_bounds__(256) void layerscale_slabs_kernel(const float* __restrict__ slabs, int nslabs, int64_t slab_stride,
                                                               const float* __restrict__ W2, const float* __restrict__ gamma,
                                                               const float* __restrict__ b2, float* __restrict__ dW2, float* __restrict__ db2,
                                                               float* __restrict__ partials, int Kdim, int Ndim, int accumulate) {
    layerscale_slabs_body(slabs, nslabs, slab_stride, W2, gamma, b2, dW2, db2, partials, Kdim, Ndim, accumulate, blockIdx.x, blockIdx.y, gridDim.y);
}

// Round 6: the layer-scale bookkeeping AND the slab sum of the block's other weight gradient (dW1 / db1 from the same pair launch,
// gemm_bf16_dma_tn_pair_kernel) as ONE launch: workgroups [0, gx * gy) are layerscale_slabs_kernel's (bx = id % gx, by = id / gx), the rest run
// reduce_rows_wide_kernel's arithmetic (same slab order, same per-element operations: bit-identical to the two launches it replaces).
struct SlabReduceJob {
    const float* partials;      // [P][pstride]
    int P;
    int64_t pstride, n, n0;
    float* out0;
    float* out1;
    float scale;
    int accumulate;
};

// Round 6, third job of the same launch: S[n] = sum_r rowscale[r / rows_per_group] * dout[r][n] -- the column sums the layer-scale / bias gradients need
// (dgamma += b2 S, db2 += gamma S) -- straight from the UNSCALED gradient and the drop-path factors, so the backward pass needs neither a scaled
// copy of dout (rowscale_kernel) nor a ones-row in the weight-gradient product.  Workgroup c sums a contiguous run of rows in row order (lanes own
// 8-column chunks and every `lanes`-th row, combined through LDS in lane order) and writes ONE partial row [b2 S_c | gamma S_c] behind the
// layer-scale rows; the fixed-order row reduction that sums those (deferred or not) adds them into dgamma and db2.
struct SRowJob {
    const bf16_t* dout;      // [M][ld] bf16 (null: no job)
    int64_t ld, M;
    const float* rowscale;      // (null: factor 1)
    int64_t rows_per_group;
    int nblocks;
};

__device__ __forceinline__ void srow_body(const SRowJob& sj, const float* __restrict__ gamma, const float* __restrict__ b2, float* __restrict__ prow,
                                          int Ndim, int c) {
    __shared__ float red2[2048];      // [lanes][N]: lanes * N = (256 / (N / 8)) * N <= 2048
    const int nch = Ndim / 8, lanes = 256 / nch;
    const int tx = threadIdx.x % nch, ty = threadIdx.x / nch;
    const int64_t rpb = (sj.M + sj.nblocks - 1) / sj.nblocks;
    const int64_t r0 = (int64_t)c * rpb, r1 = r0 + rpb < sj.M ? r0 + rpb : sj.M;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ty < lanes) {
        // four rows of a lane in flight (clamped addresses, zero factor past the end: a branch per load would serialise the round trips)
        for (int64_t r = r0 + ty; r < r1; r += 4 * lanes) {
            float v[4][8], sc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t rq = r + q * lanes, rc = rq < r1 ? rq : r1 - 1;
                load8<bf16_t>(sj.dout + rc * sj.ld + tx * 8, v[q]);
                sc[q] = rq < r1 ? (sj.rowscale ? sj.rowscale[rc / sj.rows_per_group] : 1.f) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = fmaf(v[q][u], sc[q], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) red2[ty * Ndim + tx * 8 + u] = acc[u];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Ndim; i += 256) {
        float t = red2[i];
        for (int l = 1; l < lanes; ++l) t += red2[l * Ndim + i];
        prow[i] = b2[i] * t;
        prow[Ndim + i] = gamma[i] * t;
    }
}

__global__ __launch_bounds__(256) void layerscale_slabs_reduce_kernel(const float* __restrict__ slabs, int nslabs, int64_t slab_stride,
                                                                      const float* __restrict__ W2, const float* __restrict__ gamma,
                                                                      const float* __restrict__ b2, float* __restrict__ dW2,
                                                                      float* __restrict__ db2, float* __restrict__ partials, int Kdim, int Ndim,
                                                                      int accumulate, int gx, int gy, SlabReduceJob job, int job_blocks, SRowJob sj) {
    const int id = blockIdx.x;
    const bool srow = sj.dout != nullptr;
    if (id < gx * gy) {      // (workgroup-uniform)
        layerscale_slabs_body(slabs, nslabs, slab_stride, W2, gamma, b2, dW2, db2, partials, Kdim, Ndim, accumulate, id % gx, id / gx, gy, srow);
        return;
    }
    if (id >= gx * gy + job_blocks) {
        const int c = id - gx * gy - job_blocks;
        srow_body(sj, gamma, b2, partials + (int64_t)(gy + c) * 2 * Ndim, Ndim, c);
        return;
    }
    const int64_t j = ((int64_t)(id - gx * gy) * 256 + threadIdx.x) * 4;
    if (j >= job.n) return;
    const float* __restrict__ pp = job.partials;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = 0;
    for (; p + 3 < job.P; p += 4) {      // (reduce_rows_wide_kernel's order, common.h)
        const float4 a = *reinterpret_cast<const float4*>(pp + (int64_t)p * job.pstride + j);
        const float4 b = *reinterpret_cast<const float4*>(pp + (int64_t)(p + 1) * job.pstride + j);
        const float4 c = *reinterpret_cast<const float4*>(pp + (int64_t)(p + 2) * job.pstride + j);
        const float4 d = *reinterpret_cast<const float4*>(pp + (int64_t)(p + 3) * job.pstride + j);
        s.x += (a.x + b.x) + (c.x + d.x);
        s.y += (a.y + b.y) + (c.y + d.y);
        s.z += (a.z + b.z) + (c.z + d.z);
        s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; p < job.P; ++p) {
        const float4 a = *reinterpret_cast<const float4*>(pp + (int64_t)p * job.pstride + j);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    s.x *= job.scale; s.y *= job.scale; s.z *= job.scale; s.w *= job.scale;
    float* dst = j < job.n0 ? job.out0 + j : (job.out1 ? job.out1 + (j - job.n0) : nullptr);
    if (!dst) return;
    if (job.accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(dst);
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *reinterpret_cast<float4*>(dst) = s;
}

static int layerscale_ksplits(int K) {
    int p = K / 32;
    if (p > 64) p = 64;
    if (p < 1) p = 1;
    return p;
}


// Batched 2-D transpose of bf16 matrices that live in one buffer: the K-contiguous copies of the Keras [K][N] kernels that the forward GEMMs
// read (both operands K-contiguous = the LDS-DMA pipeline of gemm_dma.h).  table[t] = {src element offset, dst element offset, rows, cols};
// block.y = matrix, block.x = 64 x 64 tile (blocks past a matrix's tile count exit).
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                                const int64_t* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
    const int64_t* e = table + 4 * blockIdx.y;
    const int64_t R = e[2], Cc = e[3];
    const int tiles_c = (int)((Cc + 63) / 64), tiles_r = (int)((R + 63) / 64);
    if ((int)blockIdx.x >= tiles_r * tiles_c) return;
    const bf16_t* s = src + e[0];
    bf16_t* d = dst + e[1];
    const int64_t r0 = (int64_t)(blockIdx.x / tiles_c) * 64, c0 = (int64_t)(blockIdx.x % tiles_c) * 64;
    const int t = threadIdx.x, tr = t >> 3, tc = (t & 7) * 8;
    const bool vec = Cc % 8 == 0 && R % 8 == 0 && ((e[0] | e[1]) % 8) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t r = r0 + tr + 32 * i, c = c0 + tc;
        bf16x8 v;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (bf16_t)0.f;
        if (r < R) {
            if (vec && c + 8 <= Cc) v = *reinterpret_cast<const bf16x8*>(s + r * Cc + c);
            else
                for (int u = 0; u < 8; ++u)
                    if (c + u < Cc) v[u] = s[r * Cc + c + u];
        }
        *reinterpret_cast<bf16x8*>(&tile[tr + 32 * i][tc]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oc = tr + 32 * i;                  // column of the source tile = row of the output
        const int64_t orow = c0 + oc, ocol = r0 + tc;
        if (orow >= Cc) continue;
        bf16x8 v;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = tile[tc + u][oc];
        if (vec && ocol + 8 <= R) *reinterpret_cast<bf16x8*>(d + orow * R + ocol) = v;
        else
            for (int u = 0; u < 8; ++u)
                if (ocol + u < R) d[orow * R + ocol + u] = v[u];
    }
}


__device__ __forceinline__ void ld4(const float* p, float* v) { Vec16<float>::load(p, v); }
__device__ __forceinline__ void ld4(const bf16_t* p, float* v) {
    const bf16x4 r = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (float)r[u];
}
__device__ __forceinline__ void st4(float* p, const float* v) { Vec16<float>::store(p, v); }
__device__ __forceinline__ void st4(bf16_t* p, const float* v) {
    bf16x4 r;
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = (bf16_t)v[u];
    *reinterpret_cast<bf16x4*>(p) = r;
}

// im2col for few input channels (stems: C = 3): with dw == 1 the KW * C elements of one (output pixel, kernel row) are ONE contiguous run in
// the NHWC image and in the patch row, so a lane moves 4 consecutive elements of a run (16-byte fp32 / 8-byte bf16 load, 8-byte bf16 / 16-byte fp32
// store) instead of one element.  Runs that touch the padding fall back to per-element bounds checks.  Requires (KW * C) % 4 == 0, ldc % 4 == 0,
// (W * C) % 4 == 0, (sw * C) % 4 == 0, (pl * C) % 4 == 0 and a 16-byte aligned image (host-checked).
template <class TI, class TO>
__global__ __launch_bounds__(256) void im2col_runs_kernel(const TI* __restrict__ x, TO* __restrict__ col, int N, int H, int W, int C, int KH,
                                                          int KW, int sh, int sw, int dh, int pt, int pl, int Ho, int Wo, int64_t ldc) {
    const int run = KW * C, q_per_run = run / 4;
    const int64_t total = (int64_t)N * Ho * Wo * KH * q_per_run;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % q_per_run);
        int64_t t = i / q_per_run;
        const int kh = (int)(t % KH);
        const int64_t m = t / KH;
        const int ow = (int)(m % Wo);
        const int oh = (int)((m / Wo) % Ho);
        const int n = (int)(m / ((int64_t)Wo * Ho));
        const int ih = oh * sh - pt + kh * dh;
        const int e0 = (ow * sw - pl) * C + 4 * q;      // element offset inside image row ih (may be negative / past the row: padding)
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ih < (unsigned)H) {
            const TI* row = x + ((int64_t)n * H + ih) * W * C;
            if (e0 >= 0 && e0 + 4 <= W * C) {
                ld4(row + e0, v);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (e0 + u >= 0 && e0 + u < W * C) v[u] = to_f32(row[e0 + u]);
            }
        }
        st4(col + m * ldc + kh * run + 4 * q, v);
    }
}

}  // namespace

extern "C" int iseg_transpose_batched(const void* src, void* dst, const int64_t* table, int count, int max_tiles, hipStream_t stream) {
    ISEG_REQUIRE(src && dst && table && count >= 0 && max_tiles >= 0, "iseg_transpose_batched: bad arguments");
    if (count == 0 || max_tiles == 0) return ISEG_OK;
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(max_tiles, count), dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, table);
    return iseg_check_launch("iseg_transpose_batched");
}

extern "C" int iseg_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, hipStream_t stream) {
    ISEG_REQUIRE(src && dst && n >= 0, "iseg_cast: bad arguments");
    if (n == 0) return ISEG_OK;
    const unsigned blocks = cap_blocks(ceil_div64(n, 8));
    if (src_dtype == ISEG_F32 && dst_dtype == ISEG_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, stream, (const float*)src, (bf16_t*)dst, n);
    else if (src_dtype == ISEG_BF16 && dst_dtype == ISEG_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, (float*)dst, n);
    else if (src_dtype == ISEG_F32 && dst_dtype == ISEG_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, stream, (const float*)src, (float*)dst, n);
    else if (src_dtype == ISEG_BF16 && dst_dtype == ISEG_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, n);
    else {
        iseg_set_error("iseg_cast: bad dtypes %d -> %d", src_dtype, dst_dtype);
        return ISEG_ERR_ARG;
    }
    return iseg_check_launch("iseg_cast");
}

extern "C" int iseg_scale_cols_cast(const float* src, const float* colscale, void* dst, int64_t rows, int cols, int dst_dtype,
                                    hipStream_t stream) {
    ISEG_REQUIRE(src && colscale && dst && rows > 0 && cols > 0, "iseg_scale_cols_cast: bad arguments");
    const unsigned blocks = cap_blocks(rows * cols);
    if (dst_dtype == ISEG_BF16)
        hipLaunchKernelGGL((scale_cols_cast_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, src, colscale, (bf16_t*)dst, rows,
                           cols);
    else
        hipLaunchKernelGGL((scale_cols_cast_kernel<float>), dim3(blocks), dim3(256), 0, stream, src, colscale, (float*)dst, rows,
                           cols);
    return iseg_check_launch("iseg_scale_cols_cast");
}

extern "C" int iseg_im2col(const void* x, int in_dtype, void* col, int out_dtype, int N, int H, int W, int C, int KH, int KW,
                           int sh, int sw, int dh, int dw, int pt, int pl, int Ho, int Wo, int64_t ldc, hipStream_t stream) {
    ISEG_REQUIRE(x && col, "iseg_im2col: null pointer");
    const int64_t K = (int64_t)KH * KW * C;
    ISEG_REQUIRE(ldc >= K, "iseg_im2col: ldc=%lld < KH*KW*C=%lld", (long long)ldc, (long long)K);
    ISEG_REQUIRE(!(in_dtype == ISEG_BF16 && out_dtype == ISEG_F32), "iseg_im2col: bf16 -> f32 unsupported");
    const int64_t M = (int64_t)N * Ho * Wo;
    const bool vec = (C % 8 == 0) && (ldc % 8 == 0);
    const bool runs = !vec && dw == 1 && (KW * C) % 4 == 0 && ldc % 4 == 0 && ((int64_t)W * C) % 4 == 0 && (sw * C) % 4 == 0 && (pl * C) % 4 == 0 &&
                      ((uintptr_t)x % 16 == 0) && ((uintptr_t)col % 16 == 0);
    const unsigned blocks = cap_blocks(runs ? M * KH * (KW * C / 4) : M * KH * KW * (vec ? C / 8 : C));
#define IM2COL(TI, TO)                                                                                                        \
    do {                                                                                                                      \
        if (runs)                                                                                                             \
            hipLaunchKernelGGL((im2col_runs_kernel<TI, TO>), dim3(blocks), dim3(256), 0, stream, (const TI*)x, (TO*)col, N, H, W, C, \
                               KH, KW, sh, sw, dh, pt, pl, Ho, Wo, ldc);                                                      \
        else if (vec)                                                                                                              \
            hipLaunchKernelGGL((im2col_kernel<TI, TO, 8>), dim3(blocks), dim3(256), 0, stream, (const TI*)x, (TO*)col, N, H, W, C, \
                               KH, KW, sh, sw, dh, dw, pt, pl, Ho, Wo, ldc);                                                  \
        else                                                                                                                  \
            hipLaunchKernelGGL((im2col_kernel<TI, TO, 1>), dim3(blocks), dim3(256), 0, stream, (const TI*)x, (TO*)col, N, H, W, C, \
                               KH, KW, sh, sw, dh, dw, pt, pl, Ho, Wo, ldc);                                                  \
        if (ldc > K)                                                                                                          \
            hipLaunchKernelGGL((zero_pad_cols_kernel<TO>), dim3(cap_blocks(M * (ldc - K))), dim3(256), 0, stream, (TO*)col, M, K, \
                               ldc);                                                                                          \
    } while (0)
    if (in_dtype == ISEG_F32 && out_dtype == ISEG_F32) IM2COL(float, float);
    else if (in_dtype == ISEG_F32 && out_dtype == ISEG_BF16) IM2COL(float, bf16_t);
    else IM2COL(bf16_t, bf16_t);
#undef IM2COL
    return iseg_check_launch("iseg_im2col");
}

extern "C" int iseg_col2im(const void* dcol, void* dx, int N, int H, int W, int C, int KH, int KW, int sh, int sw, int dh, int dw,
                           int pt, int pl, int Ho, int Wo, int64_t ldc, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(dcol && dx, "iseg_col2im: null pointer");
    const bool vec = (C % 8 == 0) && (ldc % 8 == 0);
    const unsigned blocks = cap_blocks((int64_t)N * H * W * (vec ? C / 8 : C));
#define COL2IM(T)                                                                                                              \
    do {                                                                                                                       \
        if (vec)                                                                                                               \
            hipLaunchKernelGGL((col2im_kernel<T, 8>), dim3(blocks), dim3(256), 0, stream, (const T*)dcol, (T*)dx, N, H, W, C, KH, \
                               KW, sh, sw, dh, dw, pt, pl, Ho, Wo, ldc);                                                       \
        else                                                                                                                   \
            hipLaunchKernelGGL((col2im_kernel<T, 1>), dim3(blocks), dim3(256), 0, stream, (const T*)dcol, (T*)dx, N, H, W, C, KH, \
                               KW, sh, sw, dh, dw, pt, pl, Ho, Wo, ldc);                                                       \
    } while (0)
    if (dtype == ISEG_BF16) COL2IM(bf16_t);
    else COL2IM(float);
#undef COL2IM
    return iseg_check_launch("iseg_col2im");
}

extern "C" size_t iseg_colsum_workspace_bytes(int batch, int64_t rows, int C) {
    const int V = C % 8 == 0 ? 8 : 1;
    return (size_t)batch * colsum_blocks(rows, C, V) * C * sizeof(float);
}

extern "C" int iseg_colsum(const void* x, int64_t ldx, int64_t batch_stride, int batch, int64_t rows, int C, float* out,
                           float scale, int accumulate, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(x && out && batch > 0 && rows > 0 && C > 0, "iseg_colsum: bad arguments");
    const bool vec = (C % 8 == 0) && (ldx % 8 == 0) && (batch_stride % 8 == 0) && ((uintptr_t)x % 16 == 0);
    const int V = vec ? 8 : 1;
    const int P = colsum_blocks(rows, C, V);
    const size_t need = (size_t)batch * P * C * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_colsum: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    float* const arena = batch == 1 ? iseg_deferred_partials(need, out, nullptr, accumulate, stream) : nullptr;      // (see common.h: deferred reductions)
    if (arena) ws = arena;
    const size_t lds = colsum_slab_bytes(C, V);
#define COLSUM(T, VV)                                                                                                  \
    hipLaunchKernelGGL((colsum_partial_kernel<T, VV>), dim3(P, batch), dim3(256), lds, stream, (const T*)x, ldx, batch_stride, \
                       (float*)ws, rows, C)
    if (dtype == ISEG_BF16) {
        if (vec) COLSUM(bf16_t, 8);
        else COLSUM(bf16_t, 1);
    } else {
        if (vec) COLSUM(float, 8);
        else COLSUM(float, 1);
    }
#undef COLSUM
    if (arena) iseg_deferred_push((const float*)ws, P, C, C, out, nullptr, C, scale, stream);
    else launch_reduce_rows((const float*)ws, P, C, (int64_t)P * C, batch, C, out, nullptr, C, C, scale, accumulate, stream);
    return iseg_check_launch("iseg_colsum");
}

extern "C" int iseg_broadcast_rows(const void* v, int v_dtype, void* y, int64_t ldy, int64_t batch_stride, int batch, int64_t rows,
                                   int C, float scale, int accumulate, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(v && y && C % 8 == 0 && ldy % 8 == 0 && batch_stride % 8 == 0, "iseg_broadcast_rows: bad arguments");
    const unsigned blocks = cap_blocks((int64_t)batch * rows * (C / 8));
#define BC(T, TV)                                                                                                           \
    hipLaunchKernelGGL((broadcast_rows_kernel<T, TV>), dim3(blocks), dim3(256), 0, stream, (const TV*)v, (T*)y, ldy, batch_stride, \
                       batch, rows, C, scale, accumulate)
    if (dtype == ISEG_BF16) {
        if (v_dtype == ISEG_BF16) BC(bf16_t, bf16_t);
        else BC(bf16_t, float);
    } else {
        ISEG_REQUIRE(v_dtype == ISEG_F32, "iseg_broadcast_rows: f32 output needs f32 vector");
        BC(float, float);
    }
#undef BC
    return iseg_check_launch("iseg_broadcast_rows");
}

extern "C" int iseg_axpby(const void* a, const void* b, void* y, float alpha, float beta, int64_t n, int dtype,
                          hipStream_t stream) {
    ISEG_REQUIRE(a && y && n >= 0, "iseg_axpby: bad arguments");
    if (n == 0) return ISEG_OK;
    const unsigned blocks = cap_blocks(ceil_div64(n, 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((axpby_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b,
                           (bf16_t*)y, alpha, beta, n);
    else
        hipLaunchKernelGGL((axpby_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)a, (const float*)b, (float*)y,
                           alpha, beta, n);
    return iseg_check_launch("iseg_axpby");
}

extern "C" int iseg_accumulate_pair(const float* src, int n, float* dst0, float* dst1, hipStream_t stream) {
    ISEG_REQUIRE(src && n > 0, "iseg_accumulate_pair: bad arguments");
    if (!dst0 && !dst1) return ISEG_OK;
    hipLaunchKernelGGL(accumulate_pair_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, stream, src, n, dst0, dst1);
    return iseg_check_launch("iseg_accumulate_pair");
}

extern "C" int iseg_rowscale(const void* x, const float* s, void* y, int64_t rows, int C, int64_t rows_per_group, int dtype,
                             hipStream_t stream) {
    ISEG_REQUIRE(x && s && y && C % 8 == 0 && rows_per_group > 0, "iseg_rowscale: bad arguments");
    const unsigned blocks = cap_blocks(rows * (C / 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((rowscale_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, s, (bf16_t*)y, rows, C,
                           rows_per_group);
    else
        hipLaunchKernelGGL((rowscale_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)x, s, (float*)y, rows, C,
                           rows_per_group);
    return iseg_check_launch("iseg_rowscale");
}

extern "C" int iseg_dropout(const void* x, void* y, int64_t n, float rate, uint64_t seed, const uint64_t* seed_offset, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && y && rate >= 0.f && rate < 1.f, "iseg_dropout: bad arguments");
    const unsigned blocks = cap_blocks(ceil_div64(n, 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n, rate,
                           seed, seed_offset);
    else
        hipLaunchKernelGGL((dropout_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)x, (float*)y, n, rate, seed, seed_offset);
    return iseg_check_launch("iseg_dropout");
}

extern "C" int iseg_drop_path_mask(float* s, int n, float keep_prob, uint64_t seed, const uint64_t* seed_offset, hipStream_t stream) {
    ISEG_REQUIRE(s && n > 0 && keep_prob > 0.f && keep_prob <= 1.f, "iseg_drop_path_mask: bad arguments");
    hipLaunchKernelGGL(drop_path_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, s, n, keep_prob, seed, seed_offset);
    return iseg_check_launch("iseg_drop_path_mask");
}

extern "C" int iseg_drop_path_masks(float* s, const float* keep_probs, int P, int n, uint64_t seed, const uint64_t* seed_offset, hipStream_t stream) {
    ISEG_REQUIRE(s && keep_probs && P > 0 && n > 0, "iseg_drop_path_masks: bad arguments");
    hipLaunchKernelGGL(drop_path_masks_kernel, dim3((P * n + 255) / 256), dim3(256), 0, stream, s, keep_probs, P, n, seed, seed_offset);
    return iseg_check_launch("iseg_drop_path_masks");
}

extern "C" int iseg_fill_f32(float* p, float value, int64_t n, hipStream_t stream) {
    ISEG_REQUIRE(p && n >= 0, "iseg_fill_f32: bad arguments");
    if (n == 0) return ISEG_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(cap_blocks(n)), dim3(256), 0, stream, p, value, n);
    return iseg_check_launch("iseg_fill_f32");
}

extern "C" int iseg_rsqrt_eps(const float* var, float eps, float* out, int n, hipStream_t stream) {
    ISEG_REQUIRE(var && out && n > 0, "iseg_rsqrt_eps: bad arguments");
    hipLaunchKernelGGL(rsqrt_eps_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, var, eps, out, n);
    return iseg_check_launch("iseg_rsqrt_eps");
}

extern "C" size_t iseg_layerscale_grads_workspace_bytes(int K, int N) {
    int p = K / 16;      // (the slab form's finer row strips; the tensor form uses the first layerscale_ksplits(K) rows of it)
    if (p > 128) p = 128;
    if (p < layerscale_ksplits(K)) p = layerscale_ksplits(K);
    return (size_t)(p + 512) * 2 * N * sizeof(float);      // (+ the row-sum job's <= 512 partial rows, rows twice as wide: iseg_layerscale_grads_slabs_srow)
}

extern "C" int iseg_layerscale_grads(const float* Z, const float* W2, const float* b2, const float* gamma, const float* S,
                                     float* dW2, float* dgamma, float* db2, int K, int N, int accumulate, void* ws,
                                     size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(Z && W2 && b2 && gamma && S && dW2 && dgamma && db2, "iseg_layerscale_grads: null pointer");
    const int P = layerscale_ksplits(K);
    const size_t need = (size_t)P * N * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_layerscale_grads: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    float* const arena = iseg_deferred_partials(need, dgamma, nullptr, accumulate, stream);      // (see common.h: deferred reductions)
    if (arena) ws = arena;
    hipLaunchKernelGGL(layerscale_stage1_kernel, dim3((N + 63) / 64, P), dim3(256), 0, stream, Z, W2, gamma, b2, S, dW2, db2, (float*)ws, K, N,
                       accumulate);
    if (arena) iseg_deferred_push((const float*)ws, P, N, N, dgamma, nullptr, N, 1.f, stream);
    else launch_reduce_rows((const float*)ws, P, N, 0, 1, N, dgamma, nullptr, N, 0, 1.f, accumulate, stream);
    return iseg_check_launch("iseg_layerscale_grads");
}

static int layerscale_grads_slabs_impl(const float* slabs, int nslabs, const float* W2, const float* b2, const float* gamma, float* dW2,
                                       float* dgamma, float* db2, int K, int N, int accumulate, void* ws, size_t ws_bytes,
                                       const SlabReduceJob* job, hipStream_t stream, const SRowJob* sj = nullptr);

extern "C" int iseg_layerscale_grads_slabs(const float* slabs, int nslabs, const float* W2, const float* b2, const float* gamma, float* dW2,
                                           float* dgamma, float* db2, int K, int N, int accumulate, void* ws, size_t ws_bytes,
                                           hipStream_t stream) {
    return layerscale_grads_slabs_impl(slabs, nslabs, W2, b2, gamma, dW2, dgamma, db2, K, N, accumulate, ws, ws_bytes, nullptr, stream);
}

// iseg_layerscale_grads_slabs plus, in the same launch, out0[j] / out1[j - n0] (+)= sum_p partials2[p][j] for j < n2 (the slab sum
// iseg_gemm_reduce would run for the pair launch's second product: n2 = (M + 1) N with the ones-row, n0 = M N, pstride = n2); partials2 must
// not alias `ws`.  n2, n0 multiples of 4, 16-byte aligned pointers.
extern "C" int iseg_layerscale_grads_slabs_reduce(const float* slabs, int nslabs, const float* W2, const float* b2, const float* gamma, float* dW2,
                                                  float* dgamma, float* db2, int K, int N, int accumulate, void* ws, size_t ws_bytes,
                                                  const float* partials2, int P2, int64_t n2, float* out0, float* out1, int64_t n0,
                                                  int accumulate2, hipStream_t stream) {
    ISEG_REQUIRE(partials2 && P2 >= 1 && n2 > 0 && out0 && n0 > 0 && n0 <= n2, "iseg_layerscale_grads_slabs_reduce: bad second job");
    ISEG_REQUIRE(n2 % 4 == 0 && n0 % 4 == 0 && (((uintptr_t)partials2 | (uintptr_t)out0 | (uintptr_t)out1) & 15) == 0,
                 "iseg_layerscale_grads_slabs_reduce: the second job needs n %% 4 == 0 and 16-byte aligned pointers");
    SlabReduceJob job{partials2, P2, n2, n2, n0, out0, out1, 1.f, accumulate2};
    return layerscale_grads_slabs_impl(slabs, nslabs, W2, b2, gamma, dW2, dgamma, db2, K, N, accumulate, ws, ws_bytes, &job, stream);
}

static int layerscale_grads_slabs_impl(const float* slabs, int nslabs, const float* W2, const float* b2, const float* gamma, float* dW2,
                                       float* dgamma, float* db2, int K, int N, int accumulate, void* ws, size_t ws_bytes,
                                       const SlabReduceJob* job, hipStream_t stream, const SRowJob* sj) {
    ISEG_REQUIRE(slabs && nslabs >= 1 && W2 && b2 && gamma && dW2 && dgamma && db2, "iseg_layerscale_grads_slabs: null pointer");
    ISEG_REQUIRE(N % 4 == 0 && (((uintptr_t)slabs | (uintptr_t)W2 | (uintptr_t)b2 | (uintptr_t)gamma | (uintptr_t)dW2 | (uintptr_t)db2) & 15) == 0,
                 "iseg_layerscale_grads_slabs: N %% 4 == 0 and 16-byte aligned operands");
    // (one row strip of 16 per block row where the partial buffer allows it: 13 slab loads per element are latency, more lanes hide it --
    // 1536 x 384 from 13 slabs: 17.7 us with K / 32 block rows)
    const bool srow = sj != nullptr;      // S from dout and the row factors (SRowJob): the slabs carry no ones-row, partial rows are [dgamma | db2]
    if (srow) ISEG_REQUIRE(sj->dout && sj->M > 0 && N % 8 == 0 && N <= 2048 && sj->ld % 8 == 0 && ((uintptr_t)sj->dout & 15) == 0 &&
                               (!sj->rowscale || sj->rows_per_group > 0),
                           "iseg_layerscale_grads_slabs: bad row-sum job (N %% 8, 16-byte aligned bf16 rows, rows_per_group > 0)");
    const int pw = srow ? 2 * N : N;
    // row-sum workgroups: ~64 rows each (a lane then walks three or four strides of four rows), at most 512 partial rows
    int PC = 0;
    if (srow) {
        const int64_t want = (sj->M + 63) / 64;
        PC = (int)(want < 1 ? 1 : (want > 512 ? 512 : want));
    }
    int P = K / 16;
    if ((size_t)(P + PC) * pw * sizeof(float) > ws_bytes) P = layerscale_ksplits(K);
    if (P > 128) P = 128;
    if (P < 1) P = 1;
    const size_t need = (size_t)(P + PC) * pw * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_layerscale_grads_slabs: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    float* const arena = iseg_deferred_partials(need, dgamma, srow ? db2 : nullptr, accumulate, stream);
    if (arena) ws = arena;
    const int64_t slab_stride = (int64_t)(K + (srow ? 0 : 1)) * N;
    if (job || srow) {
        const int gx = (N / 4 + 15) / 16;
        const int64_t b2n = job ? (job->n / 4 + 255) / 256 : 0;
        SlabReduceJob none{nullptr, 0, 0, 0, 0, nullptr, nullptr, 1.f, 0};
        SRowJob sr = srow ? *sj : SRowJob{nullptr, 0, 0, nullptr, 1, 0};
        sr.nblocks = PC;
        hipLaunchKernelGGL(layerscale_slabs_reduce_kernel, dim3((unsigned)(gx * P + b2n + PC)), dim3(256), 0, stream, slabs, nslabs, slab_stride, W2,
                           gamma, b2, dW2, db2, (float*)ws, K, N, accumulate, gx, P, job ? *job : none, (int)b2n, sr);
    } else {
        hipLaunchKernelGGL(layerscale_slabs_kernel, dim3((N / 4 + 15) / 16, P), dim3(256), 0, stream, slabs, nslabs, slab_stride, W2, gamma, b2,
                           dW2, db2, (float*)ws, K, N, accumulate);
    }
    if (srow) {      // (db2 is summed from the partial rows too: its direct store in the by == 0 workgroups is off)
        if (arena) iseg_deferred_push((const float*)ws, P + PC, pw, pw, dgamma, db2, N, 1.f, stream);
        else launch_reduce_rows((const float*)ws, P + PC, pw, 0, 1, pw, dgamma, db2, N, 0, 1.f, accumulate, stream);
    } else if (arena) iseg_deferred_push((const float*)ws, P, N, N, dgamma, nullptr, N, 1.f, stream);
    else launch_reduce_rows((const float*)ws, P, N, 0, 1, N, dgamma, nullptr, N, 0, 1.f, accumulate, stream);
    return iseg_check_launch("iseg_layerscale_grads_slabs");
}

// iseg_layerscale_grads_slabs_reduce whose S = colsum(rowscale * dout) comes from the unscaled gradient and the row factors (see SRowJob): the
// slabs are [nslabs][K][N] (NO ones-row); partials2 may be null (no second slab job).
extern "C" int iseg_layerscale_grads_slabs_srow(const float* slabs, int nslabs, const float* W2, const float* b2, const float* gamma, float* dW2,
                                                float* dgamma, float* db2, int K, int N, int accumulate, void* ws, size_t ws_bytes,
                                                const float* partials2, int P2, int64_t n2, float* out0, float* out1, int64_t n0, int accumulate2,
                                                const void* dout, int64_t ld_dout, int64_t M, const float* rowscale, int64_t rows_per_group,
                                                hipStream_t stream) {
    SlabReduceJob job{partials2, P2, n2, n2, n0, out0, out1, 1.f, accumulate2};
    if (partials2) {
        ISEG_REQUIRE(P2 >= 1 && n2 > 0 && out0 && n0 > 0 && n0 <= n2 && n2 % 4 == 0 && n0 % 4 == 0 &&
                         (((uintptr_t)partials2 | (uintptr_t)out0 | (uintptr_t)out1) & 15) == 0,
                     "iseg_layerscale_grads_slabs_srow: bad second job");
    }
    SRowJob sj{(const bf16_t*)dout, ld_dout, M, rowscale, rows_per_group, 0};
    return layerscale_grads_slabs_impl(slabs, nslabs, W2, b2, gamma, dW2, dgamma, db2, K, N, accumulate, ws, ws_bytes, partials2 ? &job : nullptr, stream,
                                       &sj);
}

extern "C" int iseg_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && y && (act == ISEG_ACT_RELU || act == ISEG_ACT_GELU || act == ISEG_ACT_SIGMOID || act == ISEG_ACT_SWISH), "iseg_act_fwd: bad arguments (act = %d)", act);
    if (n == 0) return ISEG_OK;
    const unsigned blocks = cap_blocks(ceil_div64(n, 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((act_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n, act);
    else
        hipLaunchKernelGGL((act_fwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)x, (float*)y, n, act);
    return iseg_check_launch("iseg_act_fwd");
}

extern "C" int iseg_act_bwd(const void* dy, const void* aux, void* dx, int64_t n, int act, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(dy && aux && dx && (act == ISEG_ACT_RELU || act == ISEG_ACT_GELU || act == ISEG_ACT_SIGMOID || act == ISEG_ACT_SWISH), "iseg_act_bwd: bad arguments (act = %d)", act);
    if (n == 0) return ISEG_OK;
    const unsigned blocks = cap_blocks(ceil_div64(n, 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((act_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)aux,
                           (bf16_t*)dx, n, act);
    else
        hipLaunchKernelGGL((act_bwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)dy, (const float*)aux,
                           (float*)dx, n, act);
    return iseg_check_launch("iseg_act_bwd");
}

extern "C" int iseg_copy2d(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols, int dtype,
                           hipStream_t stream) {
    ISEG_REQUIRE(src && dst && rows > 0 && cols > 0, "iseg_copy2d: bad arguments");
    ISEG_REQUIRE(cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "iseg_copy2d: cols/ld must be multiples of 8");
    const unsigned blocks = cap_blocks(rows * (cols / 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((copy2d_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, ld_src, (bf16_t*)dst,
                           ld_dst, rows, cols);
    else
        hipLaunchKernelGGL((copy2d_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)src, ld_src, (float*)dst,
                           ld_dst, rows, cols);
    return iseg_check_launch("iseg_copy2d");
}

extern "C" int iseg_scale_dev(const void* x, const float* s_dev, void* y, int64_t n, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && s_dev && y && n >= 0, "iseg_scale_dev: bad arguments");
    if (n == 0) return ISEG_OK;
    const unsigned blocks = cap_blocks(ceil_div64(n, 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((scale_dev_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, s_dev, (bf16_t*)y, n);
    else
        hipLaunchKernelGGL((scale_dev_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)x, s_dev, (float*)y, n);
    return iseg_check_launch("iseg_scale_dev");
}

// ---- sliding-window accumulation (core_inference.py:254-301 of the reference): fp32, arbitrary widths ----
namespace {
__global__ void add2d_f32_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd, int64_t rows,
                                 int64_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i % cols;
        dst[r * ldd + c] += src[r * lds_ + c];
    }
}
__global__ void scale_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, int64_t rows,
                                      int C) {
    const int64_t total = rows * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i] * s[i / C];
}
}  // namespace

extern "C" int iseg_add2d_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                              hipStream_t stream) {
    ISEG_REQUIRE(src && dst && rows > 0 && cols > 0, "iseg_add2d_f32: bad arguments");
    hipLaunchKernelGGL(add2d_f32_kernel, dim3(cap_blocks(rows * cols)), dim3(256), 0, stream, src, ld_src, dst, ld_dst, rows, cols);
    return iseg_check_launch("iseg_add2d_f32");
}

extern "C" int iseg_scale_rows_f32(const float* x, const float* s, float* y, int64_t rows, int C, hipStream_t stream) {
    ISEG_REQUIRE(x && s && y && rows > 0 && C > 0, "iseg_scale_rows_f32: bad arguments");
    hipLaunchKernelGGL(scale_rows_f32_kernel, dim3(cap_blocks(rows * C)), dim3(256), 0, stream, x, s, y, rows, C);
    return iseg_check_launch("iseg_scale_rows_f32");
}
