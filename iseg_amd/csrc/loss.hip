// Ignore-label softmax cross-entropy and the argmax/confusion-matrix metric, fp32, HBM-bound.
//
// Loss: losses/catecrossentropy_ignore_label.py:44-88 weighted_loss --
//   mask = y != ignore ; (ignore == 0 -> y -= 1) ; onehot(y) is the zero row for out-of-range y ;
//   loss_p = mask * w[y] * ( logsumexp(z_p) - z_p[y] )      (CategoricalCrossentropy(from_logits=True), NONE)
//   Keras then averages over ALL N*H*W positions (ignored ones included) -- the caller passes that 1/P as grad_scale.
// Metric: metrics/seg_metric_wrapper.py:89-102 + metrics/confusion_matrix.py:65-143 --
//   pred = first argmax ; ignored labels carry weight 0 ; cm[y][pred] += 1.
// A block stages PIX pixels x C logits through LDS with fully coalesced 16-B lanes, each lane then owns one
// pixel (row stride C words; conflict-free for odd C such as 21), writes its gradient row back into the same
// LDS slab, and the block streams it out coalesced again.
#include "common.h"
#include "iseg_hip.h"

namespace {

static inline int pixels_per_block(int C) {
    int pix = (48 * 1024) / (4 * C);
    pix = (pix / 64) * 64;
    if (pix > 256) pix = 256;
    if (pix < 64) pix = 64;
    return pix;
}

__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                         const float* __restrict__ class_w, int64_t P, int C, int ignore,
                                                         float* __restrict__ loss_px, float* __restrict__ block_sums,
                                                         float* __restrict__ dlogits, float grad_scale,
                                                         const float* __restrict__ grad_px, int pix, int focal, float f_alpha,
                                                         float f_gamma, unsigned long long* __restrict__ cm, int use_hist) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [pix][C], then the confusion histogram [C*C] when cm is given
    __shared__ float wsum[4];
    unsigned int* hist = reinterpret_cast<unsigned int*>(tile + (int64_t)pix * C);
    if (cm && use_hist)
        for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0u;
    // one tile per workgroup, or -- when the confusion histogram rides along -- persistent workgroups that flush it once
    const int64_t ntiles = (P + pix - 1) / pix;
    for (int64_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        __syncthreads();      // previous tile fully consumed (and the histogram zeroed on the first pass)
        const int64_t p0 = tl * pix;
        const int npx = (int)((P - p0 < pix) ? (P - p0) : pix);
        const int64_t nel = (int64_t)npx * C;
        const float* src = logits + p0 * C;
        const bool vec = ((p0 * C) % 4 == 0) && (nel % 4 == 0);
        if (vec) {
            for (int i = threadIdx.x; i < nel / 4; i += 256)
                reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(src)[i];
        } else {
            for (int i = threadIdx.x; i < nel; i += 256) tile[i] = src[i];
        }
        __syncthreads();
        float my_loss = 0.f;
        if ((int)threadIdx.x < npx) {
            float* z = tile + threadIdx.x * C;
            int y = labels[p0 + threadIdx.x];
            const bool keep = y != ignore;
            if (ignore == 0) y -= 1;
            const bool in_range = y >= 0 && y < C;
            float w = keep ? 1.f : 0.f;
            if (class_w) w *= in_range ? class_w[y] : 0.f;
            float mx = z[0];
            int best = 0;
            for (int c = 1; c < C; ++c)
                if (z[c] > mx) {      // strict: first maximal index, as tf.argmax (the running-mIoU pass rides this kernel when cm != NULL)
                    mx = z[c];
                    best = c;
                }
            if (cm && keep && in_range) {
                if (use_hist) atomicAdd(&hist[y * C + best], 1u);
                else atomicAdd(&cm[(int64_t)y * C + best], 1ull);
            }
            const float zy = in_range ? z[y] : 0.f;
            // exp(z - max) is evaluated once (v_exp_f32 path: |rel err| ~ 1e-6 on arguments in [-90, 0]) and parked in the tile for the
            // gradient; the libm expf evaluated twice per class made this kernel VALU-bound (146 us for 176 MB at cfg2)
            float se = 0.f;
            for (int c = 0; c < C; ++c) {
                const float e = __expf(z[c] - mx);
                z[c] = e;
                se += e;
            }
            const float lse = mx + __logf(se);
            if (!focal) {
                // -sum_c onehot_c * log_softmax_c : zero row when y is out of range
                my_loss = in_range ? w * (lse - zy) : 0.f;
                if (loss_px) loss_px[p0 + threadIdx.x] = my_loss;
                if (dlogits) {
                    const float g = w * grad_scale * (grad_px ? grad_px[p0 + threadIdx.x] : 1.f);
                    const float inv = in_range ? g / se : 0.f;
                    for (int c = 0; c < C; ++c) z[c] = z[c] * inv - ((c == y) ? g : 0.f);
                }
            } else {
                // keras CategoricalFocalCrossentropy(from_logits): p = clip(softmax_y, 1e-7, 1 - 1e-7),
                // loss = alpha * (1 - p)^gamma * (-log p); the clip passes no gradient outside its range
                const float py = in_range ? z[y] / se : 1.f;
                const float pc = fminf(fmaxf(py, 1e-7f), 1.f - 1e-7f);
                const float om = 1.f - pc;
                const float mod = __powf(om, f_gamma);
                const float lg = __logf(pc);
                my_loss = in_range ? w * f_alpha * mod * (-lg) : 0.f;
                if (loss_px) loss_px[p0 + threadIdx.x] = my_loss;
                if (dlogits) {
                    const float g = w * grad_scale * (grad_px ? grad_px[p0 + threadIdx.x] : 1.f);
                    const bool live = in_range && py > 1e-7f && py < 1.f - 1e-7f;
                    // dL/dp = alpha * (gamma (1-p)^(gamma-1) log p - (1-p)^gamma / p);   dp/dz_c = p (delta_cy - p_c)
                    const float dldp = live ? f_alpha * (f_gamma * (mod / om) * lg - mod / pc) : 0.f;
                    const float k = g * dldp * py;
                    const float inv = 1.f / se;
                    for (int c = 0; c < C; ++c) z[c] = k * (((c == y) ? 1.f : 0.f) - z[c] * inv);
                }
            }
        }
        if (block_sums) {
            const float s = wave_sum(my_loss);
            if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
        }
        __syncthreads();
        if (block_sums && threadIdx.x == 0) block_sums[tl] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (dlogits) {
            float* dst = dlogits + p0 * C;
            if (vec) {
                for (int i = threadIdx.x; i < nel / 4; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<float4*>(tile)[i];
            } else {
                for (int i = threadIdx.x; i < nel; i += 256) dst[i] = tile[i];
            }
        }
    }
    __syncthreads();
    if (cm && use_hist)
        for (int i = threadIdx.x; i < C * C; i += 256)
            if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);      // integer counts: order-independent, exact
}

__global__ void sum_blocks_kernel(const float* __restrict__ v, int n, float* __restrict__ out, float scale) {
    // single block, fixed-order tree -> deterministic
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// persistent blocks walk the pixel tiles; labels/predictions are binned in an LDS histogram (C*C counters, int atomics are
// order-independent, so the result is exact and deterministic) that is flushed to the global uint64 matrix once per block
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                               int64_t P, int C, int ignore, int32_t* __restrict__ pred_out,
                                                               unsigned long long* __restrict__ cm, int pix, int use_hist) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [pix][C] floats, then int hist[C*C]
    unsigned int* hist = reinterpret_cast<unsigned int*>(tile + (int64_t)pix * C);
    if (use_hist)
        for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0u;
    const int64_t ntiles = (P + pix - 1) / pix;
    for (int64_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int64_t p0 = tl * pix;
        const int npx = (int)((P - p0 < pix) ? (P - p0) : pix);
        const int64_t nel = (int64_t)npx * C;
        const float* src = logits + p0 * C;
        const bool vec = ((p0 * C) % 4 == 0) && (nel % 4 == 0);
        __syncthreads();  // previous tile fully consumed (and hist zeroed on the first pass)
        if (vec) {
            for (int i = threadIdx.x; i < nel / 4; i += 256)
                reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(src)[i];
        } else {
            for (int i = threadIdx.x; i < nel; i += 256) tile[i] = src[i];
        }
        __syncthreads();
        if ((int)threadIdx.x < npx) {
            const float* z = tile + threadIdx.x * C;
            int best = 0;
            float bv = z[0];
            for (int c = 1; c < C; ++c)
                if (z[c] > bv) {  // strict: first maximal index, as tf.argmax
                    bv = z[c];
                    best = c;
                }
            if (pred_out) pred_out[p0 + threadIdx.x] = best;
            if (cm && labels) {
                const int y = labels[p0 + threadIdx.x];
                if (y != ignore && y >= 0 && y < C) {
                    if (use_hist) atomicAdd(&hist[y * C + best], 1u);
                    else atomicAdd(&cm[(int64_t)y * C + best], 1ull);
                }
            }
        }
    }
    __syncthreads();
    if (use_hist && cm)
        for (int i = threadIdx.x; i < C * C; i += 256)
            if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}

}  // namespace

extern "C" size_t iseg_softmax_ce_workspace_bytes(int64_t P, int C) {
    return (size_t)ceil_div64(P, pixels_per_block(C)) * sizeof(float);
}

static int launch_softmax_ce(const float* logits, const int32_t* labels, const float* class_w, int64_t P, int C,
                                      int ignore_label, float* loss_px, float* loss_sum, float loss_sum_scale, float* dlogits,
                                      float grad_scale, const float* grad_px, void* ws, size_t ws_bytes,
                                      hipStream_t stream, int focal, float f_alpha, float f_gamma, unsigned long long* cm, const char* who) {
    ISEG_REQUIRE(logits && labels && P > 0 && C > 0, "%s: bad arguments", who);
    ISEG_REQUIRE(C <= 640, "%s: num_class %d > 640 unsupported", who, C);
    const int pix = pixels_per_block(C);
    const int64_t blocks = ceil_div64(P, pix);
    float* bs = nullptr;
    if (loss_sum) {
        const size_t need = (size_t)blocks * sizeof(float);
        if (!ws || ws_bytes < need) {
            iseg_set_error("%s: needs %zu workspace bytes, got %zu", who, need, ws_bytes);
            return ISEG_ERR_WORKSPACE;
        }
        bs = (float*)ws;
    }
    const int use_hist = (cm != nullptr) && ((size_t)C * C * 4 <= 16 * 1024);
    const size_t lds = (size_t)pix * C * sizeof(float) + (use_hist ? (size_t)C * C * 4 : 0);
    const int64_t grid = (cm && blocks > 2048) ? 2048 : blocks;      // histogram flushes: one per workgroup
    hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)grid), dim3(256), lds, stream, logits, labels, class_w, P, C,
                       ignore_label, loss_px, bs, dlogits, grad_scale, grad_px, pix, focal, f_alpha, f_gamma, cm, use_hist);
    if (loss_sum)
        hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(256), 0, stream, (const float*)bs, (int)blocks, loss_sum,
                           loss_sum_scale);
    return iseg_check_launch(who);
}

extern "C" int iseg_softmax_ce_ignore(const float* logits, const int32_t* labels, const float* class_w, int64_t P, int C,
                                      int ignore_label, float* loss_px, float* loss_sum, float loss_sum_scale, float* dlogits,
                                      float grad_scale, const float* grad_px, void* ws, size_t ws_bytes,
                                      hipStream_t stream) {
    return launch_softmax_ce(logits, labels, class_w, P, C, ignore_label, loss_px, loss_sum, loss_sum_scale, dlogits, grad_scale, grad_px,
                             ws, ws_bytes, stream, 0, 0.f, 0.f, nullptr, "iseg_softmax_ce_ignore");
}

extern "C" int iseg_softmax_ce_confusion(const float* logits, const int32_t* labels, const float* class_w, int64_t P, int C,
                                        int ignore_label, float* loss_px, float* loss_sum, float loss_sum_scale, float* dlogits,
                                        float grad_scale, const float* grad_px, uint64_t* cm, void* ws, size_t ws_bytes,
                                        hipStream_t stream) {
    ISEG_REQUIRE(cm, "iseg_softmax_ce_confusion: null confusion matrix");
    return launch_softmax_ce(logits, labels, class_w, P, C, ignore_label, loss_px, loss_sum, loss_sum_scale, dlogits, grad_scale, grad_px,
                             ws, ws_bytes, stream, 0, 0.f, 0.f, (unsigned long long*)cm, "iseg_softmax_ce_confusion");
}

extern "C" int iseg_softmax_focal_ce_ignore(const float* logits, const int32_t* labels, const float* class_w, int64_t P, int C,
                                            int ignore_label, float alpha, float gamma, float* loss_px, float* loss_sum,
                                            float loss_sum_scale, float* dlogits, float grad_scale, const float* grad_px, void* ws,
                                            size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(gamma >= 0.f, "iseg_softmax_focal_ce_ignore: gamma must be >= 0");
    return launch_softmax_ce(logits, labels, class_w, P, C, ignore_label, loss_px, loss_sum, loss_sum_scale, dlogits, grad_scale, grad_px,
                             ws, ws_bytes, stream, 1, alpha, gamma, nullptr, "iseg_softmax_focal_ce_ignore");
}

extern "C" int iseg_argmax_confusion(const float* logits, const int32_t* labels, int64_t P, int C, int ignore_label,
                                     int32_t* pred_out, unsigned long long* cm, hipStream_t stream) {
    ISEG_REQUIRE(logits && P > 0 && C > 0 && C <= 640, "iseg_argmax_confusion: bad arguments");
    ISEG_REQUIRE(!cm || labels, "iseg_argmax_confusion: confusion matrix needs labels");
    const int pix = pixels_per_block(C);
    int64_t blocks = ceil_div64(P, pix);
    if (blocks > 2048) blocks = 2048;
    const int use_hist = (cm != nullptr) && ((size_t)C * C * 4 <= 16 * 1024);
    const size_t lds = (size_t)pix * C * sizeof(float) + (use_hist ? (size_t)C * C * 4 : 0);
    hipLaunchKernelGGL(argmax_confusion_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, logits, labels, P, C, ignore_label,
                       pred_out, cm, pix, use_hist);
    return iseg_check_launch("iseg_argmax_confusion");
}
