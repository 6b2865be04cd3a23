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

// Gradient clipping as Keras' base optimizer applies it (optimizer.py _clip_gradients; the reference passes clipnorm / clipvalue
// straight through, core_optimizer.py:170-183), after AdamW_EXT's NaN scrub (adamw.py:63-74):
//   clipnorm        per VARIABLE: g * clipnorm / max(||g||_2, clipnorm)                      (tf.clip_by_norm)
//   global_clipnorm all variables: g * clipnorm * min(1 / ||all g||_2, 1 / clipnorm)        (tf.clip_by_global_norm)
//   clipvalue       per element
// The squared norms come from iseg_grad_sqnorm: fixed-order sums (256-element blocks, then the blocks of a variable, then the
// variables), so the step stays deterministic.
struct ClipArgs {
    const float* seg_sq;      // [nseg] sum of squares of the scrubbed, scaled gradient of each variable, or NULL
    float clipnorm;           // <= 0: off
    const float* global_sq;   // [1] sum over all variables, or NULL
    float global_clipnorm;    // <= 0: off
};
__device__ __forceinline__ float clip_factor(const ClipArgs& c, int seg) {
    float f = 1.f;
    if (c.seg_sq && c.clipnorm > 0.f) f = c.clipnorm / fmaxf(sqrtf(c.seg_sq[seg]), c.clipnorm);
    else if (c.global_sq && c.global_clipnorm > 0.f) f = c.global_clipnorm * fminf(1.f / sqrtf(c.global_sq[0]), 1.f / c.global_clipnorm);
    return f;
}

// hp[0]=lr, hp[1]=sqrt(1-b2^t)/(1-b1^t), hp[2]=grad_scale (e.g. 1/world), hp[3]=clipvalue (<=0: off)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float* __restrict__ vhat, bf16_t* __restrict__ w_bf16,
                                                    const int32_t* __restrict__ seg_of_block, const float* __restrict__ seg_lr_mult,
                                                    const float* __restrict__ seg_wd, const float* __restrict__ hp, float b1,
                                                    float b2, float eps, int64_t nblocks, ClipArgs clip) {
    const float lr = hp[0], corr = hp[1], gscale = hp[2], clipv = hp[3];
    // one wavefront per 256-element block (the segment granularity of the flat parameter buffer), four elements per lane: 16-byte
    // loads / stores, the segment lookup is wave-uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < nblocks; b += (int64_t)gridDim.x * 4) {
        const int seg = seg_of_block[b];
        if (seg < 0) continue;  // padding block
        const float lr_mult = seg_lr_mult[seg], wd = seg_wd[seg];
        // a variable this optimizer does not own (MultiOptimizer routes it to another one) or a frozen one: no update, no decay -- and its
        // gradient is not even read: NaN * 0 from another group's variable must not reach the weight through this pass
        if (lr_mult == 0.f && wd == 0.f) continue;
        const float cf = clip_factor(clip, seg);
        const int64_t i = b * 256 + lane * 4;
        float4 w4 = *reinterpret_cast<const float4*>(w + i);
        const float4 g4 = *reinterpret_cast<const float4*>(g + i);
        float4 m4 = *reinterpret_cast<const float4*>(m + i);
        float4 v4 = *reinterpret_cast<const float4*>(v + i);
        float4 h4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vhat) h4 = *reinterpret_cast<const float4*>(vhat + i);
        float wi[4] = {w4.x, w4.y, w4.z, w4.w}, mi[4] = {m4.x, m4.y, m4.z, m4.w}, vi[4] = {v4.x, v4.y, v4.z, v4.w};
        float hi[4] = {h4.x, h4.y, h4.z, h4.w};
        const float gr[4] = {g4.x, g4.y, g4.z, g4.w};
        const float alpha = lr * lr_mult * corr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float gi = scrub_nan(gr[u]) * gscale * cf;
            if (clipv > 0.f) gi = fminf(fmaxf(gi, -clipv), clipv);
            wi[u] -= wi[u] * wd * lr;
            mi[u] += (gi - mi[u]) * (1.f - b1);
            vi[u] += (gi * gi - vi[u]) * (1.f - b2);
            float den = vi[u];
            if (vhat) {      // amsgrad (adamw.py:54-57): v_hat = max(v_hat, v) replaces v in the denominator
                hi[u] = fmaxf(hi[u], vi[u]);
                den = hi[u];
            }
            wi[u] -= (mi[u] * alpha) / (sqrtf(den) + eps);
        }
        *reinterpret_cast<float4*>(w + i) = make_float4(wi[0], wi[1], wi[2], wi[3]);
        *reinterpret_cast<float4*>(m + i) = make_float4(mi[0], mi[1], mi[2], mi[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vi[0], vi[1], vi[2], vi[3]);
        if (vhat) *reinterpret_cast<float4*>(vhat + i) = make_float4(hi[0], hi[1], hi[2], hi[3]);
        if (w_bf16) {
            bf16x4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = (bf16_t)wi[u];
            *reinterpret_cast<bf16x4*>(w_bf16 + i) = o;
        }
    }
}

// hp[0]=lr, hp[2]=grad_scale, hp[3]=clipvalue ; l2 adds 2*l2*w to the gradient (keras l2 regularizer of set_weight_decay).
// SGD_EXT does not scrub NaN gradients (only AdamW_EXT overrides _clip_gradients), so neither does this kernel.
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                  bf16_t* __restrict__ w_bf16, const int32_t* __restrict__ seg_of_block,
                                                  const float* __restrict__ seg_lr_mult, const float* __restrict__ seg_l2,
                                                  const float* __restrict__ hp, float momentum, int nesterov, int64_t nblocks,
                                                  ClipArgs clip) {
    const float lr = hp[0], gscale = hp[2], clipv = hp[3];
    for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int seg = seg_of_block[b];
        if (seg < 0) continue;
        const float lr_mult = seg_lr_mult[seg], l2 = seg_l2[seg];
        if (lr_mult == 0.f && l2 == 0.f) continue;      // not this optimizer's variable (see adamw_kernel): never touch it, never read its gradient
        const int64_t i = b * 256 + threadIdx.x;
        float wi = w[i];
        float gi = (g[i] * gscale + 2.f * l2 * wi) * clip_factor(clip, seg);
        if (clipv > 0.f) gi = fminf(fmaxf(gi, -clipv), clipv);
        const float mi = -gi * lr * lr_mult + m[i] * momentum;
        wi += nesterov ? (-gi * lr * lr_mult + mi * momentum) : mi;      // sgd.py:46-49
        w[i] = wi;
        m[i] = mi;
        if (w_bf16) w_bf16[i] = (bf16_t)wi;
    }
}

// ---- squared gradient norms for clipnorm / global_clipnorm: three fixed-order stages ----
__global__ __launch_bounds__(256) void grad_sq_blocks_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                             const int32_t* __restrict__ seg_of_block, const float* __restrict__ seg_l2,
                                                             const float* __restrict__ hp, int scrub, float* __restrict__ block_sq,
                                                             int64_t nblocks) {
    const float gscale = hp[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < nblocks; b += (int64_t)gridDim.x * 4) {
        const int seg = seg_of_block[b];
        float s = 0.f;
        if (seg >= 0) {
            const int64_t i = b * 256 + lane * 4;
            const float4 g4 = *reinterpret_cast<const float4*>(g + i);
            float gr[4] = {g4.x, g4.y, g4.z, g4.w};
            float wr[4] = {0.f, 0.f, 0.f, 0.f};
            const float l2 = seg_l2 ? seg_l2[seg] : 0.f;
            if (l2 != 0.f) {
                const float4 w4 = *reinterpret_cast<const float4*>(w + i);
                wr[0] = w4.x, wr[1] = w4.y, wr[2] = w4.z, wr[3] = w4.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float gi = (scrub ? scrub_nan(gr[u]) : gr[u]) * gscale + 2.f * l2 * wr[u];
                s = fmaf(gi, gi, s);
            }
        }
        s = wave_sum(s);
        if (lane == 0) block_sq[b] = s;
    }
}
// one workgroup per variable: its blocks are contiguous; strided partials + fixed-order tree
__global__ __launch_bounds__(256) void grad_sq_segments_kernel(const float* __restrict__ block_sq, const int32_t* __restrict__ seg_first_block,
                                                               float* __restrict__ seg_sq) {
    __shared__ float red[256];
    const int seg = blockIdx.x;
    const int b0 = seg_first_block[seg], b1 = seg_first_block[seg + 1];
    float s = 0.f;
    for (int b = b0 + threadIdx.x; b < b1; b += 256) s += block_sq[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) seg_sq[seg] = red[0];
}
__global__ __launch_bounds__(256) void grad_sq_total_kernel(const float* __restrict__ seg_sq, int nseg, float* __restrict__ total) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nseg; i += 256) s += seg_sq[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = red[0];
}

}  // namespace

extern "C" int iseg_grad_sqnorm(const float* g, const float* w, const int32_t* seg_of_block, const int32_t* seg_first_block, const float* seg_l2,
                                const float* hp, int scrub_nan_grads, float* block_ws, float* seg_sq, float* global_sq, int64_t nblocks,
                                int nseg, hipStream_t stream) {
    ISEG_REQUIRE(g && seg_of_block && seg_first_block && hp && block_ws && seg_sq && nblocks > 0 && nseg > 0, "iseg_grad_sqnorm: bad arguments");
    ISEG_REQUIRE(!seg_l2 || w, "iseg_grad_sqnorm: the l2 term needs the weights");
    const unsigned grid = (unsigned)(nblocks < 256 * 16 ? nblocks : 256 * 16);
    hipLaunchKernelGGL(grad_sq_blocks_kernel, dim3(grid), dim3(256), 0, stream, g, w, seg_of_block, seg_l2, hp, scrub_nan_grads, block_ws, nblocks);
    hipLaunchKernelGGL(grad_sq_segments_kernel, dim3(nseg), dim3(256), 0, stream, (const float*)block_ws, seg_first_block, seg_sq);
    if (global_sq) hipLaunchKernelGGL(grad_sq_total_kernel, dim3(1), dim3(256), 0, stream, (const float*)seg_sq, nseg, global_sq);
    return iseg_check_launch("iseg_grad_sqnorm");
}

extern "C" int iseg_adamw_step(float* w, const float* g, float* m, float* v, float* vhat, void* w_bf16, const int32_t* seg_of_block,
                               const float* seg_lr_mult, const float* seg_wd, const float* hp, float beta1, float beta2, float eps,
                               const float* seg_sq, float clipnorm, const float* global_sq, float global_clipnorm, int64_t nblocks,
                               hipStream_t stream) {
    ISEG_REQUIRE(w && g && m && v && seg_of_block && seg_lr_mult && seg_wd && hp && nblocks > 0, "iseg_adamw_step: bad arguments");
    ISEG_REQUIRE(!(clipnorm > 0.f) || seg_sq, "iseg_adamw_step: clipnorm needs the per-variable squared norms (iseg_grad_sqnorm)");
    ISEG_REQUIRE(!(global_clipnorm > 0.f) || global_sq, "iseg_adamw_step: global_clipnorm needs the global squared norm (iseg_grad_sqnorm)");
    const unsigned grid = (unsigned)(nblocks < 256 * 16 ? nblocks : 256 * 16);
    const ClipArgs clip{seg_sq, clipnorm, global_sq, global_clipnorm};
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, stream, w, g, m, v, vhat, (bf16_t*)w_bf16, seg_of_block, seg_lr_mult, seg_wd,
                       hp, beta1, beta2, eps, nblocks, clip);
    return iseg_check_launch("iseg_adamw_step");
}

extern "C" int iseg_sgd_momentum_step(float* w, const float* g, float* m, void* w_bf16, const int32_t* seg_of_block,
                                      const float* seg_lr_mult, const float* seg_l2, const float* hp, float momentum, int nesterov,
                                      const float* seg_sq, float clipnorm, const float* global_sq, float global_clipnorm,
                                      int64_t nblocks, hipStream_t stream) {
    ISEG_REQUIRE(w && g && m && seg_of_block && seg_lr_mult && seg_l2 && hp && nblocks > 0, "iseg_sgd_momentum_step: bad arguments");
    ISEG_REQUIRE(!(clipnorm > 0.f) || seg_sq, "iseg_sgd_momentum_step: clipnorm needs the per-variable squared norms (iseg_grad_sqnorm)");
    ISEG_REQUIRE(!(global_clipnorm > 0.f) || global_sq, "iseg_sgd_momentum_step: global_clipnorm needs the global squared norm");
    const unsigned grid = (unsigned)(nblocks < 256 * 16 ? nblocks : 256 * 16);
    const ClipArgs clip{seg_sq, clipnorm, global_sq, global_clipnorm};
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, stream, w, g, m, (bf16_t*)w_bf16, seg_of_block, seg_lr_mult, seg_l2, hp,
                       momentum, nesterov, nblocks, clip);
    return iseg_check_launch("iseg_sgd_momentum_step");
}
