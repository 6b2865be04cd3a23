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


// ---------------------------------------------------------------------------------------------------------------------------------
// Fused logits tail of the training step: bilinear upsample (tf.image.resize, half-pixel centres, TF's lerp order) of the low-resolution
// logits + ignore-label cross-entropy + its gradient folded back through the resize + argmax / confusion matrix, WITHOUT the
// [N, Ho, Wo, C] fp32 logits and gradient tensors (88 MB each at 16 x 512 x 512 x 21, four HBM passes in the materialised route:
// layers/core_model_ext.py:199-256 compute_logits_upsample / compute_final_results, losses/catecrossentropy_ignore_label.py:44-88,
// metrics/seg_metric_wrapper.py:89-102).
//
// For an integer upsampling factor (sy even, sx a power of two <= 64) the output pixels whose two source rows are (b - 1, b) form the
// band y in [b sy - sy/2, b sy + sy/2), and likewise segments of sx columns.  A wavefront owns one band x 64 columns: a lane keeps its
// column, so the x-interpolated source rows top[c], bottom[c] - top[c] are computed once and a pixel costs one fma per class; it walks
// the band's rows accumulating the row-weighted gradient for the two source rows (a0, a1), reduces over the sx lanes of each segment
// with xor-shuffles, and writes one partial [2 rows][2 columns][C] per (band, segment).  A second tiny kernel adds the <= 4 x 4
// partials that meet in a source cell in a fixed order (deterministic; no atomics on floats).  Counts go through an LDS histogram.
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lerp_src(int dst, float scale, int in_size, int& lo, int& hi, float& t) {      // = resize.hip lerp_of
    const float src = ((float)dst + 0.5f) * scale - 0.5f;
    const float f = floorf(src);
    lo = max((int)f, 0);
    hi = min((int)ceilf(src), in_size - 1);
    t = src - f;
}

// CMAX: register arrays; EXACT: C == CMAX known at compile time (the common 19 / 21 class heads: no per-class bound checks, no padding).
// A band is cut into `nsplit` row ranges, one wavefront each, so that the grid is several resident rounds deep (even tail).
template <class TI, int CMAX, bool EXACT>
__global__ __launch_bounds__(256) void upsample_ce_kernel(const TI* __restrict__ z, const int32_t* __restrict__ labels,
                                                          const float* __restrict__ class_w, int N, int Hi, int Wi, int Ho, int Wo, int Crt,
                                                          int sy, int sx, int nwc, int nsplit, int ignore, float grad_scale,
                                                          float* __restrict__ item_loss, float* __restrict__ partial,
                                                          unsigned long long* __restrict__ cm) {
    const int C = EXACT ? CMAX : Crt;
    __shared__ unsigned int hist[CMAX * CMAX];
    if (cm)
        for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nitems = N * (Hi + 1) * nsplit * nwc;
    const int item = blockIdx.x * 4 + wid;
    if (item < nitems) {
        const int wc = item % nwc, part = (item / nwc) % nsplit, b = (item / (nwc * nsplit)) % (Hi + 1), n = item / (nwc * nsplit * (Hi + 1));
        const float fy = (float)Hi / (float)Ho, fx = (float)Wi / (float)Wo;
        const int x = 64 * wc - sx / 2 + lane;
        const bool xv = x >= 0 && x < Wo;
        int clo, chi;
        float tx;
        lerp_src(xv ? x : 0, fx, Wi, clo, chi, tx);
        const int rlo = max(b - 1, 0), rhi = min(b, Hi - 1);
        const int rows_per_part = (sy + nsplit - 1) / nsplit;
        const int yb0 = b * sy - sy / 2 + part * rows_per_part;
        const int y0 = max(yb0, 0), y1 = min(min(yb0 + rows_per_part, b * sy + sy / 2), Ho);
        float top[CMAX], dif[CMAX], a0[CMAX], a1[CMAX];
        {
            const TI* r0 = z + ((int64_t)(n * Hi + rlo) * Wi) * C;
            const TI* r1 = z + ((int64_t)(n * Hi + rhi) * Wi) * C;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                top[c] = dif[c] = a0[c] = a1[c] = 0.f;
                if (c < C) {
                    const float tl = to_f32(r0[clo * C + c]), tr = to_f32(r0[chi * C + c]);
                    const float bl = to_f32(r1[clo * C + c]), br = to_f32(r1[chi * C + c]);
                    const float tp = tl + (tr - tl) * tx;
                    const float bt = bl + (br - bl) * tx;
                    top[c] = tp;
                    dif[c] = bt - tp;
                }
            }
        }
        float loss = 0.f;
        const int32_t* lab = labels + ((int64_t)n * Ho) * Wo + (xv ? x : 0);
        constexpr int RB = 8;      // rows whose labels are requested together (one dependent L2 round trip per row otherwise)
        for (int yb = y0; yb < y1; yb += RB) {
            int labs[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) labs[q] = (xv && yb + q < y1) ? lab[(int64_t)(yb + q) * Wo] : ignore;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int y = yb + q;
                if (y >= y1) break;
                int ylo, yhi;
                float t;
                lerp_src(y, fy, Hi, ylo, yhi, t);
                int yl = labs[q];
                const bool keep = xv && yl != ignore;
                if (ignore == 0) yl -= 1;
                const bool in_range = yl >= 0 && yl < C;
                float w = keep ? 1.f : 0.f;
                if (class_w) w *= in_range ? class_w[yl] : 0.f;
                float v[CMAX];
                float mx = -3.0e38f;
                int best = 0;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    v[c] = top[c] + dif[c] * t;
                    if (c < C && v[c] > mx) {      // strict: first maximal index, as tf.argmax
                        mx = v[c];
                        best = c;
                    }
                }
                if (cm && keep && in_range) atomicAdd(&hist[yl * C + best], 1u);
                float se = 0.f, zy = 0.f;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    if (c == yl) zy = v[c];
                    v[c] = c < C ? __expf(v[c] - mx) : 0.f;
                    se += v[c];
                }
                const float lse = mx + __logf(se);
                loss += in_range ? w * (lse - zy) : 0.f;
                if (partial) {
                    const float g = w * grad_scale;
                    const float inv = in_range ? g / se : 0.f;
                    const float w1 = t, w0 = 1.f - t;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        const float d = v[c] * inv - ((c == yl) ? g : 0.f);
                        a0[c] = fmaf(w0, d, a0[c]);
                        a1[c] = fmaf(w1, d, a1[c]);
                    }
                }
            }
        }
        loss = wave_sum(loss);
        if (lane == 0) item_loss[item] = loss;
        if (partial) {
            // [row lo / hi][column lo / hi][c], summed over the sx lanes of this lane's segment (xor butterfly: every lane ends with the sum)
            const int cseg = (64 * wc + lane) / sx;
            float* dst = partial + ((((int64_t)(n * (Hi + 1) + b) * nsplit + part) * (Wi + 1) + cseg) * 4) * C;
            const float wx1 = xv ? tx : 0.f, wx0 = xv ? 1.f - tx : 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                float q00 = wx0 * a0[c], q01 = wx1 * a0[c], q10 = wx0 * a1[c], q11 = wx1 * a1[c];
                for (int o = sx >> 1; o > 0; o >>= 1) {
                    q00 += __shfl_xor(q00, o, 64);
                    q01 += __shfl_xor(q01, o, 64);
                    q10 += __shfl_xor(q10, o, 64);
                    q11 += __shfl_xor(q11, o, 64);
                }
                if (c < C && (lane & (sx - 1)) == 0 && cseg <= Wi) {
                    dst[c] = q00;
                    dst[C + c] = q01;
                    dst[2 * C + c] = q10;
                    dst[3 * C + c] = q11;
                }
            }
        }
    }
    __syncthreads();
    if (cm)
        for (int i = threadIdx.x; i < C * C; i += 256)
            if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}

// dz[n, i, j, c] = the partials of the (band, row-slot) x (segment, column-slot) pairs that address source cell (i, j), in a fixed order
template <class TI>
__global__ void upsample_ce_gather_kernel(const float* __restrict__ partial, TI* __restrict__ dz, int N, int Hi, int Wi, int C, int nsplit) {
    const int64_t total = (int64_t)N * Hi * Wi * C;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int j = (int)((e / C) % Wi), i = (int)((e / ((int64_t)C * Wi)) % Hi), n = (int)(e / ((int64_t)C * Wi * Hi));
        // (band, slot) pairs whose source row is i: (i, hi) and (i + 1, lo) always; the clamped edge bands add (0, lo) and (Hi, hi)
        int bb[4], ys[4], nb = 0;
        bb[nb] = i, ys[nb++] = 1;
        bb[nb] = i + 1, ys[nb++] = 0;
        if (i == 0) bb[nb] = 0, ys[nb++] = 0;
        if (i == Hi - 1) bb[nb] = Hi, ys[nb++] = 1;
        int cc[4], xs[4], nc = 0;
        cc[nc] = j, xs[nc++] = 1;
        cc[nc] = j + 1, xs[nc++] = 0;
        if (j == 0) cc[nc] = 0, xs[nc++] = 0;
        if (j == Wi - 1) cc[nc] = Wi, xs[nc++] = 1;
        float s = 0.f;
        for (int u = 0; u < nb; ++u)
            for (int q = 0; q < nsplit; ++q)
                for (int v = 0; v < nc; ++v)
                    s += partial[(((((int64_t)(n * (Hi + 1) + bb[u]) * nsplit + q) * (Wi + 1) + cc[v]) * 4) + ys[u] * 2 + xs[v]) * C + c];
        dz[e] = from_f32<TI>(s);
    }
}

static inline int upsample_ce_nsplit(int N, int Hi, int nwc, int sy) {
    // measured at 16 x 512 x 512 / 21 classes (2448 band x column items): 128 us unsplit, 139 us with 4 row ranges per band -- the
    // per-wavefront setup (84 source-logit loads, the 84-value segment reduction) outweighs the evener tail.  The kernel is VALU-bound
    // (SQ_ACTIVE_INST_VALU = 36 % of wave cycles at 4.2 cycles per instruction, ~480 instructions per pixel row): split only small grids
    int ns = 1;
    while (ns < 4 && (int64_t)N * (Hi + 1) * nwc * ns < 1024 && sy / (ns * 2) >= 8) ns *= 2;
    return ns;
}

static inline bool upsample_ce_geometry(int Hi, int Wi, int Ho, int Wo, int C, int& sy, int& sx) {
    if (Hi <= 0 || Wi <= 0 || Ho % Hi != 0 || Wo % Wi != 0 || C <= 0 || C > 32) return false;
    sy = Ho / Hi;
    sx = Wo / Wi;
    return sy >= 2 && sy % 2 == 0 && sx >= 2 && sx <= 64 && (sx & (sx - 1)) == 0;
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
    ISEG_REQUIRE(C <= 256, "%s: num_class %d > 256 unsupported (one 64-pixel tile of fp32 logits must fit 64 KiB of LDS)", who, C);
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
    ISEG_REQUIRE(logits && P > 0 && C > 0 && C <= 256, "iseg_argmax_confusion: bad arguments (num_class <= 256)");
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

extern "C" int iseg_upsample_ce_supported(int Hi, int Wi, int Ho, int Wo, int C) {
    int sy, sx;
    return upsample_ce_geometry(Hi, Wi, Ho, Wo, C, sy, sx) ? 1 : 0;
}

extern "C" size_t iseg_upsample_ce_workspace_bytes(int N, int Hi, int Wi, int Ho, int Wo, int C) {
    int sy, sx;
    if (!upsample_ce_geometry(Hi, Wi, Ho, Wo, C, sy, sx)) return 0;
    const int nwc = ((Wi + 1) * sx + 63) / 64;
    const int ns = upsample_ce_nsplit(N, Hi, nwc, sy);
    const size_t items = (size_t)N * (Hi + 1) * ns * nwc;
    return (items + 3) / 4 * 4 * sizeof(float) + (size_t)N * (Hi + 1) * ns * (Wi + 1) * 4 * C * sizeof(float);
}

extern "C" int iseg_upsample_ce(const void* z, int dtype, const int32_t* labels, const float* class_w, int N, int Hi, int Wi, int Ho, int Wo,
                                int C, int ignore_label, float* loss_sum, float loss_sum_scale, void* dz, float grad_scale, uint64_t* cm,
                                void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(z && labels && loss_sum && N > 0, "iseg_upsample_ce: bad arguments");
    ISEG_REQUIRE(dtype == ISEG_F32 || dtype == ISEG_BF16, "iseg_upsample_ce: bad dtype %d", dtype);
    int sy, sx;
    ISEG_REQUIRE(upsample_ce_geometry(Hi, Wi, Ho, Wo, C, sy, sx),
                 "iseg_upsample_ce: needs an integer upsampling factor (rows: even; columns: a power of two <= 64) and num_class <= 32; got "
                 "%dx%d -> %dx%d, %d classes (use iseg_resize_bilinear_fwd + iseg_softmax_ce_ignore)", Hi, Wi, Ho, Wo, C);
    ISEG_REQUIRE((int64_t)N * Ho * Wo < (1ll << 31), "iseg_upsample_ce: too many pixels");
    const size_t need = iseg_upsample_ce_workspace_bytes(N, Hi, Wi, Ho, Wo, C);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_upsample_ce: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const int nwc = ((Wi + 1) * sx + 63) / 64;
    const int ns = upsample_ce_nsplit(N, Hi, nwc, sy);
    const int items = N * (Hi + 1) * ns * nwc;
    float* item_loss = (float*)ws;
    float* partial = dz ? item_loss + (items + 3) / 4 * 4 : nullptr;
    const int blocks = (items + 3) / 4;
#define UCE(TI, CMAX, EXACT)                                                                                                            \
    hipLaunchKernelGGL((upsample_ce_kernel<TI, CMAX, EXACT>), dim3(blocks), dim3(256), 0, stream, (const TI*)z, labels, class_w, N, Hi, Wi, Ho, \
                       Wo, C, sy, sx, nwc, ns, ignore_label, grad_scale, item_loss, partial, (unsigned long long*)cm)
#define UCE_T(TI)                          \
    do {                                   \
        if (C == 21) UCE(TI, 21, true);    \
        else if (C == 19) UCE(TI, 19, true); \
        else if (C <= 8) UCE(TI, 8, false);  \
        else if (C <= 24) UCE(TI, 24, false); \
        else UCE(TI, 32, false);             \
    } while (0)
    if (dtype == ISEG_BF16) UCE_T(bf16_t);
    else UCE_T(float);
#undef UCE_T
#undef UCE
    hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(256), 0, stream, (const float*)item_loss, items, loss_sum, loss_sum_scale);
    if (dz) {
        const int64_t total = (int64_t)N * Hi * Wi * C;
        const int gb = (int)(ceil_div64(total, 256) < 1024 ? ceil_div64(total, 256) : 1024);
        if (dtype == ISEG_BF16)
            hipLaunchKernelGGL((upsample_ce_gather_kernel<bf16_t>), dim3(gb), dim3(256), 0, stream, (const float*)partial, (bf16_t*)dz, N, Hi, Wi, C, ns);
        else
            hipLaunchKernelGGL((upsample_ce_gather_kernel<float>), dim3(gb), dim3(256), 0, stream, (const float*)partial, (float*)dz, N, Hi, Wi, C, ns);
    }
    return iseg_check_launch("iseg_upsample_ce");
}
