// tf.image.resize(method=bilinear / nearest, half-pixel centres, no antialias) on NHWC tensors, as called by
// utils/common.py:107-134 resize_image (logits -> input size in layers/core_model_ext.py:217-226; FPN top-down
// path layers/fpn.py:40-61; multi-scale inference core_model.py:170-229).
//   src = (dst + 0.5) * in/out - 0.5 ; lo = max(floor(src),0) ; hi = min(ceil(src), in-1) ; t = src - floor(src)
//   out = top + (bottom - top)*ty with top = tl + (tr - tl)*tx            (lerp order kept: it fixes rounding)
// HBM-bound: the forward is one coalesced write of the output; the backward is the exact transpose in gather
// form, separated into an X pass and a Y pass so that no atomics are needed and the result is deterministic.
#include "common.h"
#include "iseg_hip.h"

namespace {

struct Lerp {
    int lo, hi;
    float t;
};

__device__ __forceinline__ Lerp lerp_of(int dst, float scale, int in_size) {
    const float src = ((float)dst + 0.5f) * scale - 0.5f;
    const float f = floorf(src);
    Lerp l;
    l.lo = max((int)f, 0);
    l.hi = min((int)ceilf(src), in_size - 1);
    l.t = src - f;
    return l;
}

template <class TI, class TO>
__global__ void resize_bilinear_fwd_kernel(const TI* __restrict__ x, TO* __restrict__ y, int N, int Hi, int Wi, int Ho, int Wo,
                                           int C, float sy, float sx) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t i0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i0 < total; i0 += (int64_t)gridDim.x * blockDim.x * 4) {
        float out[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u;
            out[u] = 0.f;
            if (i < total) {
                const int c = (int)(i % C);
                int64_t r = i / C;
                const int ox = (int)(r % Wo);
                r /= Wo;
                const int oy = (int)(r % Ho);
                const int n = (int)(r / Ho);
                const Lerp ly = lerp_of(oy, sy, Hi), lx = lerp_of(ox, sx, Wi);
                const TI* base = x + (int64_t)n * Hi * Wi * C + c;
                const float tl = to_f32(base[((int64_t)ly.lo * Wi + lx.lo) * C]);
                const float tr = to_f32(base[((int64_t)ly.lo * Wi + lx.hi) * C]);
                const float bl = to_f32(base[((int64_t)ly.hi * Wi + lx.lo) * C]);
                const float br = to_f32(base[((int64_t)ly.hi * Wi + lx.hi) * C]);
                const float top = tl + (tr - tl) * lx.t;
                const float bot = bl + (br - bl) * lx.t;
                out[u] = top + (bot - top) * ly.t;
            }
        }
        if (i0 + 4 <= total && (total % 4 == 0)) {
            if (sizeof(TO) == 4) {
                *reinterpret_cast<float4*>(y + i0) = make_float4(out[0], out[1], out[2], out[3]);
            } else {
                bf16x4 v;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (bf16_t)out[u];
                *reinterpret_cast<bf16x4*>(y + i0) = v;
            }
        } else {
            for (int u = 0; u < 4 && i0 + u < total; ++u) y[i0 + u] = from_f32<TO>(out[u]);
        }
    }
}

// one axis of the transposed interpolation: out[o, j, q] = sum_{d in D(j)} w(d -> j) * in[o, d, q]
//   outer o (size O), reduced axis d (size Dn, "destination" of the forward), kept axis j (size J, forward source),
//   inner q (size Q, contiguous).   scale = J / Dn (forward in/out ratio along this axis)
template <class TI, class TO>
__global__ void resize_bwd_axis_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t O, int Dn, int J, int64_t Q,
                                       float scale, const TO* __restrict__ add) {
    const int64_t total = O * J * Q;
    const float inv = 1.0f / scale;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i % Q;
        const int j = (int)((i / Q) % J);
        const int64_t o = i / (Q * J);
        // destinations whose lo or hi can equal j: src in (j-1, j+1)  ->  dst in ((j-0.5)*inv-0.5 , (j+1.5)*inv-0.5)
        int d0 = (int)floorf(((float)j - 0.5f) * inv - 0.5f) - 1;
        int d1 = (int)ceilf(((float)j + 1.5f) * inv - 0.5f) + 1;
        if (j == 0) d0 = 0;             // clamped sources: everything below maps to lo = hi = 0
        if (j == J - 1) d1 = Dn - 1;    // and everything above to J-1
        d0 = max(d0, 0);
        d1 = min(d1, Dn - 1);
        float acc = 0.f;
        const TI* p = in + o * Dn * Q + q;
        for (int d = d0; d <= d1; ++d) {
            const Lerp l = lerp_of(d, scale, J);
            float w = 0.f;
            if (l.lo == j) w += 1.f - l.t;
            if (l.hi == j) w += l.t;
            if (w != 0.f) acc += w * to_f32(p[(int64_t)d * Q]);
        }
        if (add) acc += to_f32(add[i]);
        out[i] = from_f32<TO>(acc);
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
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const unsigned blocks = cap_blocks(ceil_div64((int64_t)N * Ho * Wo * C, 4));
#define RS(TI, TO)                                                                                                            \
    hipLaunchKernelGGL((resize_bilinear_fwd_kernel<TI, TO>), dim3(blocks), dim3(256), 0, stream, (const TI*)x, (TO*)y, N, Hi, Wi, \
                       Ho, Wo, C, sy, sx)
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
    // X pass: [N*Ho, Wo, C] -> [N*Ho, Wi, C]
    const int64_t t1 = (int64_t)N * Ho * Wi * C;
    if (dy_dtype == ISEG_BF16)
        hipLaunchKernelGGL((resize_bwd_axis_kernel<bf16_t, float>), dim3(cap_blocks(t1)), dim3(256), 0, stream, (const bf16_t*)dy,
                           tmp, (int64_t)N * Ho, Wo, Wi, (int64_t)C, sx, (const float*)nullptr);
    else
        hipLaunchKernelGGL((resize_bwd_axis_kernel<float, float>), dim3(cap_blocks(t1)), dim3(256), 0, stream, (const float*)dy, tmp,
                           (int64_t)N * Ho, Wo, Wi, (int64_t)C, sx, (const float*)nullptr);
    // Y pass: [N, Ho, Wi*C] -> [N, Hi, Wi*C]
    const int64_t t2 = (int64_t)N * Hi * Wi * C;
    if (dx_dtype == ISEG_BF16)
        hipLaunchKernelGGL((resize_bwd_axis_kernel<float, bf16_t>), dim3(cap_blocks(t2)), dim3(256), 0, stream, (const float*)tmp,
                           (bf16_t*)dx, (int64_t)N, Ho, Hi, (int64_t)Wi * C, sy, (const bf16_t*)dx_add);
    else
        hipLaunchKernelGGL((resize_bwd_axis_kernel<float, float>), dim3(cap_blocks(t2)), dim3(256), 0, stream, (const float*)tmp,
                           (float*)dx, (int64_t)N, Ho, Hi, (int64_t)Wi * C, sy, (const float*)dx_add);
    return iseg_check_launch("iseg_resize_bilinear_bwd");
}

extern "C" int iseg_resize_nearest_i32(const int32_t* x, int32_t* y, int N, int Hi, int Wi, int Ho, int Wo, int C,
                                       hipStream_t stream) {
    ISEG_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "iseg_resize_nearest_i32: bad arguments");
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    hipLaunchKernelGGL(resize_nearest_i32_kernel, dim3(cap_blocks((int64_t)N * Ho * Wo * C)), dim3(256), 0, stream, x, y, N, Hi, Wi,
                       Ho, Wo, C, sy, sx);
    return iseg_check_launch("iseg_resize_nearest_i32");
}
