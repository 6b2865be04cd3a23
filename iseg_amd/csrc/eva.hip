// EVA-02 operators that the rest of the library has no kernel for (backbones/eva/* of the reference):
//   * rotary position embedding on the packed [q | k | v] rows of an attention layer (backbones/eva/attention.py:136-146,
//     backbones/eva/rotar_embedding_cat.py:117-135 apply_rot_embed_cat / rot), together with the q / v bias of the fused projection
//     (attention.py:100-112: bias = [q_bias | 0 | v_bias]) -- ONE in-place pass over the [B, T, 3C] tensor, so q and k never make a round trip as
//     separate [B, heads, T, d] tensors;
//   * the gated product of the SwiGLU / GluMlp blocks, out = act(gate) * x (backbones/eva/swiglu.py:88-92, glumlp.py:96-103), forward and backward,
//     on strided operands (GluMlp's gate and value are the two column halves of one Dense output).
// Both are single HBM passes (memory-bound): 16-byte vectors, grid-stride.
#include "common.h"
#include "iseg_hip.h"

namespace {

// One thread = 8 consecutive columns of one row of qkv [rows = B T][3 C].  Columns [0, C) = q, [C, 2C) = k, [2C, 3C) = v; inside q / k the head
// dimension index is col % hd, and the rotation pairs (2 i, 2 i + 1) never straddle an 8-column piece.  emb [T - prefix][2 hd] fp32 = [sin | cos]
// (rotar_embedding_cat.py:137-171: both halves hold every band twice, so sin[2 i] == sin[2 i + 1]; the kernel reads both entries as the reference does).
// forward (inverse = 0):  x' = x cos + rot(x) sin,  rot(x)[2 i] = -x[2 i + 1], rot(x)[2 i + 1] = x[2 i]     (bias added first)
// backward (inverse = 1): the transposed map  dx[2 i] = d[2 i] cos[2 i] + d[2 i + 1] sin[2 i + 1],  dx[2 i + 1] = d[2 i + 1] cos[2 i + 1] - d[2 i] sin[2 i]
template <class T>
__global__ __launch_bounds__(256) void qkv_rope_kernel(const T* src, T* dst, const float* __restrict__ q_bias, const float* __restrict__ v_bias,
                                                       const float* __restrict__ emb, int64_t rows, int Tn, int prefix, int C, int hd, int inverse) {
    const int pieces = 3 * C / 8;
    const int64_t total = rows * pieces;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / pieces;
        const int col = (int)(i - r * pieces) * 8;
        const int part = col / C, c = col - part * C;
        const int t = (int)(r % Tn);
        const float* bias = part == 0 ? q_bias : (part == 2 ? v_bias : nullptr);
        const bool rotate = part < 2 && t >= prefix && emb != nullptr;
        if (!bias && !rotate && src == dst) continue;      // (in place: nothing to do for this piece; out of place: it is copied)
        float v[8];
        load8<T>(src + r * 3 * C + col, v);
        if (bias) {
            float b[8];
            load8<float>(bias + c, b);
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] += b[u];
        }
        if (rotate) {
            const int d = c % hd;
            float sn[8], cs[8];
            const float* e = emb + (int64_t)(t - prefix) * 2 * hd + d;
            load8<float>(e, sn);
            load8<float>(e + hd, cs);
            float o[8];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float x0 = v[2 * p], x1 = v[2 * p + 1];
                if (!inverse) {
                    o[2 * p] = x0 * cs[2 * p] - x1 * sn[2 * p];
                    o[2 * p + 1] = x1 * cs[2 * p + 1] + x0 * sn[2 * p + 1];
                } else {
                    o[2 * p] = x0 * cs[2 * p] + x1 * sn[2 * p + 1];
                    o[2 * p + 1] = x1 * cs[2 * p + 1] - x0 * sn[2 * p];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = o[u];
        }
        store8<T>(dst + r * 3 * C + col, v);
    }
}

// activation of the gate: GELU (exact erf in fp32 storage, the polynomial forms of common.h in bf16), swish / silu, sigmoid
template <bool FAST> __device__ __forceinline__ void glu_act(float g, int act, float& a, float& da) {
    if (act == ISEG_ACT_GELU) {
        if (FAST) gelu_sig_both(g, a, da);
        else {
            a = gelu_erf(g);
            da = gelu_erf_grad(g);
        }
    } else {
        const float s = 1.f / (1.f + expf(-g));
        if (act == ISEG_ACT_SWISH) {
            a = g * s;
            da = s * (1.f + g * (1.f - s));
        } else {
            a = s;
            da = s * (1.f - s);
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void glu_fwd_kernel(const T* __restrict__ g, int64_t ldg, const T* __restrict__ x, int64_t ldx, T* __restrict__ out,
                                                      int64_t ldo, int64_t rows, int cols, int act) {
    const int cv = cols / 8;
    const int64_t total = rows * cv;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cv;
        const int c = (int)(i - r * cv) * 8;
        float gv[8], xv[8];
        load8<T>(g + r * ldg + c, gv);
        load8<T>(x + r * ldx + c, xv);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a, da;
            glu_act<sizeof(T) == 2>(gv[u], act, a, da);
            xv[u] *= a;
        }
        store8<T>(out + r * ldo + c, xv);
    }
}

// dg = dout x act'(g),  dx = dout act(g)
template <class T>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const T* __restrict__ dout, int64_t ldd, const T* __restrict__ g, int64_t ldg,
                                                      const T* __restrict__ x, int64_t ldx, T* __restrict__ dg, int64_t ldgg, T* __restrict__ dx,
                                                      int64_t ldxg, int64_t rows, int cols, int act) {
    const int cv = cols / 8;
    const int64_t total = rows * cv;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cv;
        const int c = (int)(i - r * cv) * 8;
        float dv[8], gv[8], xv[8], og[8], ox[8];
        load8<T>(dout + r * ldd + c, dv);
        load8<T>(g + r * ldg + c, gv);
        load8<T>(x + r * ldx + c, xv);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a, da;
            glu_act<sizeof(T) == 2>(gv[u], act, a, da);
            og[u] = dv[u] * xv[u] * da;
            ox[u] = dv[u] * a;
        }
        store8<T>(dg + r * ldgg + c, og);
        store8<T>(dx + r * ldxg + c, ox);
    }
}

// any width / any row pitch (EVA02-large: int(1024 * 8 / 3) = 2730 hidden units, rows 4-byte aligned only): one element per lane and trip
template <class T>
__global__ __launch_bounds__(256) void glu_fwd_any_kernel(const T* __restrict__ g, int64_t ldg, const T* __restrict__ x, int64_t ldx, T* __restrict__ out,
                                                          int64_t ldo, int64_t rows, int cols, int act) {
    const int64_t total = rows * cols;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        float a, da;
        glu_act<sizeof(T) == 2>(to_f32(g[r * ldg + c]), act, a, da);
        out[r * ldo + c] = from_f32<T>(to_f32(x[r * ldx + c]) * a);
    }
}

template <class T>
__global__ __launch_bounds__(256) void glu_bwd_any_kernel(const T* __restrict__ dout, int64_t ldd, const T* __restrict__ g, int64_t ldg,
                                                          const T* __restrict__ x, int64_t ldx, T* __restrict__ dg, int64_t ldgg, T* __restrict__ dx,
                                                          int64_t ldxg, int64_t rows, int cols, int act) {
    const int64_t total = rows * cols;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        float a, da;
        glu_act<sizeof(T) == 2>(to_f32(g[r * ldg + c]), act, a, da);
        const float d = to_f32(dout[r * ldd + c]);
        dg[r * ldgg + c] = from_f32<T>(d * to_f32(x[r * ldx + c]) * da);
        dx[r * ldxg + c] = from_f32<T>(d * a);
    }
}

unsigned eva_blocks(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int iseg_qkv_rope(const void* qkv, void* out, const float* q_bias, const float* v_bias, const float* emb, int64_t rows, int tokens, int prefix, int C,
                             int head_dim, int inverse, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(qkv && out && rows > 0 && tokens > 0 && rows % tokens == 0 && prefix >= 0 && prefix <= tokens, "iseg_qkv_rope: bad arguments");
    ISEG_REQUIRE(C > 0 && head_dim > 0 && C % head_dim == 0 && head_dim % 8 == 0, "iseg_qkv_rope: C = %d, head_dim = %d (head_dim must be a multiple of 8)", C,
                 head_dim);
    ISEG_REQUIRE((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)q_bias | (uintptr_t)v_bias | (uintptr_t)emb) & 15) == 0, "iseg_qkv_rope: operands must be 16-byte aligned");
    if (!q_bias && !v_bias && !emb && qkv == out) return ISEG_OK;
    const unsigned blocks = eva_blocks(rows * (3 * C / 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((qkv_rope_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)qkv, (bf16_t*)out, q_bias, v_bias, emb, rows, tokens, prefix, C, head_dim,
                           inverse);
    else
        hipLaunchKernelGGL((qkv_rope_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)qkv, (float*)out, q_bias, v_bias, emb, rows, tokens, prefix, C, head_dim,
                           inverse);
    return iseg_check_launch("iseg_qkv_rope");
}

static bool glu_act_ok(int act) { return act == ISEG_ACT_GELU || act == ISEG_ACT_SWISH || act == ISEG_ACT_SIGMOID; }

extern "C" int iseg_glu_fwd(const void* gate, int64_t ld_gate, const void* x, int64_t ld_x, void* out, int64_t ld_out, int64_t rows, int cols, int act,
                            int dtype, hipStream_t stream) {
    ISEG_REQUIRE(gate && x && out && rows > 0 && cols > 0 && glu_act_ok(act), "iseg_glu_fwd: bad arguments (act = %d)", act);
    if (cols % 8 || ld_gate % 8 || ld_x % 8 || ld_out % 8 || (((uintptr_t)gate | (uintptr_t)x | (uintptr_t)out) & 15)) {      // no 16-byte pieces: the scalar form
        const unsigned nb = eva_blocks(rows * cols);
        if (dtype == ISEG_BF16)
            hipLaunchKernelGGL((glu_fwd_any_kernel<bf16_t>), dim3(nb), dim3(256), 0, stream, (const bf16_t*)gate, ld_gate, (const bf16_t*)x, ld_x, (bf16_t*)out,
                               ld_out, rows, cols, act);
        else
            hipLaunchKernelGGL((glu_fwd_any_kernel<float>), dim3(nb), dim3(256), 0, stream, (const float*)gate, ld_gate, (const float*)x, ld_x, (float*)out, ld_out,
                               rows, cols, act);
        return iseg_check_launch("iseg_glu_fwd");
    }
    const unsigned blocks = eva_blocks(rows * (cols / 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((glu_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)gate, ld_gate, (const bf16_t*)x, ld_x, (bf16_t*)out,
                           ld_out, rows, cols, act);
    else
        hipLaunchKernelGGL((glu_fwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)gate, ld_gate, (const float*)x, ld_x, (float*)out, ld_out,
                           rows, cols, act);
    return iseg_check_launch("iseg_glu_fwd");
}

extern "C" int iseg_glu_bwd(const void* dout, int64_t ld_dout, const void* gate, int64_t ld_gate, const void* x, int64_t ld_x, void* dgate,
                            int64_t ld_dgate, void* dx, int64_t ld_dx, int64_t rows, int cols, int act, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(dout && gate && x && dgate && dx && rows > 0 && cols > 0 && glu_act_ok(act), "iseg_glu_bwd: bad arguments (act = %d)", act);
    if (cols % 8 || ld_dout % 8 || ld_gate % 8 || ld_x % 8 || ld_dgate % 8 || ld_dx % 8 ||
        (((uintptr_t)dout | (uintptr_t)gate | (uintptr_t)x | (uintptr_t)dgate | (uintptr_t)dx) & 15)) {      // no 16-byte pieces: the scalar form
        const unsigned nb = eva_blocks(rows * cols);
        if (dtype == ISEG_BF16)
            hipLaunchKernelGGL((glu_bwd_any_kernel<bf16_t>), dim3(nb), dim3(256), 0, stream, (const bf16_t*)dout, ld_dout, (const bf16_t*)gate, ld_gate,
                               (const bf16_t*)x, ld_x, (bf16_t*)dgate, ld_dgate, (bf16_t*)dx, ld_dx, rows, cols, act);
        else
            hipLaunchKernelGGL((glu_bwd_any_kernel<float>), dim3(nb), dim3(256), 0, stream, (const float*)dout, ld_dout, (const float*)gate, ld_gate,
                               (const float*)x, ld_x, (float*)dgate, ld_dgate, (float*)dx, ld_dx, rows, cols, act);
        return iseg_check_launch("iseg_glu_bwd");
    }
    const unsigned blocks = eva_blocks(rows * (cols / 8));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((glu_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dout, ld_dout, (const bf16_t*)gate, ld_gate,
                           (const bf16_t*)x, ld_x, (bf16_t*)dgate, ld_dgate, (bf16_t*)dx, ld_dx, rows, cols, act);
    else
        hipLaunchKernelGGL((glu_bwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)dout, ld_dout, (const float*)gate, ld_gate, (const float*)x,
                           ld_x, (float*)dgate, ld_dgate, (float*)dx, ld_dx, rows, cols, act);
    return iseg_check_launch("iseg_glu_bwd");
}
