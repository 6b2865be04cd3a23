// Fused optimizer steps over the flat parameter buffer (HBM-bound, ~16-18 B per parameter).
//
// optimizers/modern/adamw.py:13-74 (AdamW_EXT.update_step / _clip_gradients) on top of Keras' AdamW:
//   g <- NaN -> 0 ; decoupled decay  w -= w * wd * lr        (Keras base optimizer, skipped for excluded names,
//                                                             utils/train_utils.py:8-37; lr WITHOUT lr_multiplier)
//   m += (g - m)(1-b1) ; v += (g*g - v)(1-b2) ; alpha = lr*lr_mult*sqrt(1-b2^t)/(1-b1^t) ; w -= m*alpha/(sqrt(v)+eps)
// optimizers/modern/sgd.py:12-51 (SGD_EXT): m = -g*lr*lr_mult + m*mu ; w += m.
//
// All parameters live in ONE flat fp32 buffer (each tensor padded to a multiple of 256 elements), so the whole
// model is one launch; a 256-element block looks up its tensor's (lr_mult, wd) in a per-block segment table.
// Step-dependent scalars (lr, bias corrections) are read from device memory so a captured hipGraph replays
// with fresh values.  The kernel also emits the bf16 shadow copy of the weights used by the MFMA kernels.
#include "common.h"
#include "iseg_hip.h"

namespace {

__device__ __forceinline__ float scrub_nan(float g) { return (g != g) ? 0.f : g; }

// hp[0]=lr, hp[1]=sqrt(1-b2^t)/(1-b1^t), hp[2]=grad_scale (e.g. 1/world), hp[3]=clipvalue (<=0: off)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ w_bf16,
                                                    const int32_t* __restrict__ seg_of_block, const float* __restrict__ seg_lr_mult,
                                                    const float* __restrict__ seg_wd, const float* __restrict__ hp, float b1,
                                                    float b2, float eps, int64_t nblocks) {
    const float lr = hp[0], corr = hp[1], gscale = hp[2], clipv = hp[3];
    // one wavefront per 256-element block (the segment granularity of the flat parameter buffer), four elements per lane: 16-byte
    // loads / stores, the segment lookup is wave-uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < nblocks; b += (int64_t)gridDim.x * 4) {
        const int seg = seg_of_block[b];
        if (seg < 0) continue;  // padding block
        const float lr_mult = seg_lr_mult[seg], wd = seg_wd[seg];
        const int64_t i = b * 256 + lane * 4;
        float4 w4 = *reinterpret_cast<const float4*>(w + i);
        const float4 g4 = *reinterpret_cast<const float4*>(g + i);
        float4 m4 = *reinterpret_cast<const float4*>(m + i);
        float4 v4 = *reinterpret_cast<const float4*>(v + i);
        float wi[4] = {w4.x, w4.y, w4.z, w4.w}, mi[4] = {m4.x, m4.y, m4.z, m4.w}, vi[4] = {v4.x, v4.y, v4.z, v4.w};
        const float gr[4] = {g4.x, g4.y, g4.z, g4.w};
        const float alpha = lr * lr_mult * corr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float gi = scrub_nan(gr[u]) * gscale;
            if (clipv > 0.f) gi = fminf(fmaxf(gi, -clipv), clipv);
            wi[u] -= wi[u] * wd * lr;
            mi[u] += (gi - mi[u]) * (1.f - b1);
            vi[u] += (gi * gi - vi[u]) * (1.f - b2);
            wi[u] -= (mi[u] * alpha) / (sqrtf(vi[u]) + eps);
        }
        *reinterpret_cast<float4*>(w + i) = make_float4(wi[0], wi[1], wi[2], wi[3]);
        *reinterpret_cast<float4*>(m + i) = make_float4(mi[0], mi[1], mi[2], mi[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vi[0], vi[1], vi[2], vi[3]);
        if (w_bf16) {
            bf16x4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = (bf16_t)wi[u];
            *reinterpret_cast<bf16x4*>(w_bf16 + i) = o;
        }
    }
}

// hp[0]=lr, hp[2]=grad_scale, hp[3]=clipvalue ; l2 adds 2*l2*w to the gradient (keras l2 regularizer of set_weight_decay)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                  bf16_t* __restrict__ w_bf16, const int32_t* __restrict__ seg_of_block,
                                                  const float* __restrict__ seg_lr_mult, const float* __restrict__ seg_l2,
                                                  const float* __restrict__ hp, float momentum, int64_t nblocks) {
    const float lr = hp[0], gscale = hp[2], clipv = hp[3];
    for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int seg = seg_of_block[b];
        if (seg < 0) continue;
        const float lr_mult = seg_lr_mult[seg], l2 = seg_l2[seg];
        const int64_t i = b * 256 + threadIdx.x;
        float wi = w[i];
        float gi = scrub_nan(g[i]) * gscale + 2.f * l2 * wi;
        if (clipv > 0.f) gi = fminf(fmaxf(gi, -clipv), clipv);
        const float mi = -gi * lr * lr_mult + m[i] * momentum;
        wi += mi;
        w[i] = wi;
        m[i] = mi;
        if (w_bf16) w_bf16[i] = (bf16_t)wi;
    }
}

}  // namespace

extern "C" int iseg_adamw_step(float* w, const float* g, float* m, float* v, void* w_bf16, const int32_t* seg_of_block,
                               const float* seg_lr_mult, const float* seg_wd, const float* hp, float beta1, float beta2, float eps,
                               int64_t nblocks, hipStream_t stream) {
    ISEG_REQUIRE(w && g && m && v && seg_of_block && seg_lr_mult && seg_wd && hp && nblocks > 0, "iseg_adamw_step: bad arguments");
    const unsigned grid = (unsigned)(nblocks < 256 * 16 ? nblocks : 256 * 16);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, stream, w, g, m, v, (bf16_t*)w_bf16, seg_of_block, seg_lr_mult, seg_wd,
                       hp, beta1, beta2, eps, nblocks);
    return iseg_check_launch("iseg_adamw_step");
}

extern "C" int iseg_sgd_momentum_step(float* w, const float* g, float* m, void* w_bf16, const int32_t* seg_of_block,
                                      const float* seg_lr_mult, const float* seg_l2, const float* hp, float momentum,
                                      int64_t nblocks, hipStream_t stream) {
    ISEG_REQUIRE(w && g && m && seg_of_block && seg_lr_mult && seg_l2 && hp && nblocks > 0, "iseg_sgd_momentum_step: bad arguments");
    const unsigned grid = (unsigned)(nblocks < 256 * 16 ? nblocks : 256 * 16);
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, stream, w, g, m, (bf16_t*)w_bf16, seg_of_block, seg_lr_mult, seg_l2, hp,
                       momentum, nblocks);
    return iseg_check_launch("iseg_sgd_momentum_step");
}
