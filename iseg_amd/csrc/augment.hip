// On-device input pipeline: the training augmentations of data_process/pipeline.py:85-170 (StandardAugmentationsPipeline) of the reference
// -- RandomScaleAugment (utils.py:303-370: bilinear image / nearest label resize by one of the discrete factors), PadAugment
// (augments/pad_augment.py: bottom / right padding with the mean pixel / ignore label), RandomCropAugment, RandomFlipAugment,
// RandomErasingAugment (noise fill, label -> ignore) -- and the input normalisation of data_process/input_norm.py:7-80, composed into ONE
// gather per output pixel: the crop window is sampled straight from the source image through the inverse of (resize -> pad -> crop -> flip),
// so none of the intermediate images exists.  The random decisions are drawn on the host (per sample: scale -> new size, crop offset, flip,
// erase rectangles) and arrive as a small int table; only the erase noise is drawn on the device (counter-based, per pixel and channel).
// HBM-bound: reads <= 4 source pixels and writes 12 B + 4 B per output pixel.
#include "common.h"
#include "iseg_hip.h"

namespace {

constexpr int AUG_INTS = 28;      // H, W, newH, newW, off_y, off_x, flip, n_erase, then 5 x (y, x, h, w)
// optional photometric parameters per sample (round 3; NULL table = none): brightness delta (0 = not executed), contrast factor (1), the channel
// means of the scaled + brightness-adjusted image (tf.image.adjust_contrast's reference point), saturation factor (1), hue delta (0),
// stddev of the evaluation noise (0)
constexpr int AUG_FLOATS = 8;

__device__ __forceinline__ float aug_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

template <class TI> __device__ __forceinline__ float px(const TI* p) { return (float)*p; }

__device__ __forceinline__ float clip256(float v) { return fminf(fmaxf(v, 0.f), 256.f); }

// tf.image.adjust_saturation / adjust_hue on float images (core/kernels/image/adjust_{saturation,hue}_op: RGB -> HSV, S * factor clipped to
// [0, 1] / H + delta wrapped to [0, 1), HSV -> RGB); the conversion is scale-free, so it runs on the [0, 256] pixel scale as it is
__device__ __forceinline__ void saturation_hue(float* v, float sat, float dh) {
    const float r = v[0], g = v[1], b = v[2];
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b)), range = mx - mn;
    float h = 0.f;
    if (range > 0.f) {
        const float norm = 1.f / (6.f * range);
        if (r == mx) h = norm * (g - b);
        else if (g == mx) h = norm * (b - r) + 2.f / 6.f;
        else h = norm * (r - g) + 4.f / 6.f;
        if (h < 0.f) h += 1.f;
    }
    float s = mx > 0.f ? range / mx : 0.f;
    if (sat != 1.f) s = fminf(fmaxf(s * sat, 0.f), 1.f);      // (adjust_hue alone never clips S: contrast may have pushed a channel below 0)
    h += dh;
    h -= floorf(h);
    const float c = s * mx, m = mx - c;
    const float dhh = h * 6.f, fm = fmodf(dhh, 2.f), x = c * (1.f - fabsf(fm - 1.f));
    const int hc = (int)dhh;
    float rr = 0.f, gg = 0.f, bb = 0.f;
    switch (hc) {
        case 0: rr = c; gg = x; break;
        case 1: rr = x; gg = c; break;
        case 2: gg = c; bb = x; break;
        case 3: gg = x; bb = c; break;
        case 4: rr = x; bb = c; break;
        default: rr = c; bb = x; break;
    }
    v[0] = rr + m;
    v[1] = gg + m;
    v[2] = bb + m;
}

__device__ __forceinline__ float aug_normal(uint64_t seed, uint64_t idx) {      // Box-Muller on two counter-based uniforms
    const float u1 = fmaxf(aug_uniform(seed, 2 * idx), 5.9604645e-8f), u2 = aug_uniform(seed, 2 * idx + 1);
    return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}

struct AugConst {
    float mean[3], a[3], b[3];      // pad colour; out = v * a + b
};

// one bilinear sample of the randomly scaled image at (ry, rx) of sample b (tf.image.resize, half-pixel centres)
template <class TI>
__device__ __forceinline__ void scaled_pixel(const TI* __restrict__ src, int Ws, int H, int W, int nH, int nW, int ry, int rx, float* v) {
    const float fy = (float)H / (float)nH, fx = (float)W / (float)nW;
    const float sy = ((float)ry + 0.5f) * fy - 0.5f, sx = ((float)rx + 0.5f) * fx - 0.5f;
    const float y0f = floorf(sy), x0f = floorf(sx);
    const int y0 = max((int)y0f, 0), y1 = min((int)ceilf(sy), H - 1), x0 = max((int)x0f, 0), x1 = min((int)ceilf(sx), W - 1);
    const float ty = sy - y0f, tx = sx - x0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float tl = px(src + ((int64_t)y0 * Ws + x0) * 3 + c), tr = px(src + ((int64_t)y0 * Ws + x1) * 3 + c);
        const float bl = px(src + ((int64_t)y1 * Ws + x0) * 3 + c), br = px(src + ((int64_t)y1 * Ws + x1) * 3 + c);
        const float tp = tl + (tr - tl) * tx, bt = bl + (br - bl) * tx;
        v[c] = tp + (bt - tp) * ty;
    }
}

// per-sample channel sums of the scaled, brightness-adjusted image: partial sums per (sample, block) in a fixed order, finished on the host side
// of the call by a second tiny launch -- the reference point of tf.image.adjust_contrast (mean over H, W of the image it is given)
template <class TI>
__global__ __launch_bounds__(256) void augment_means_kernel(const TI* __restrict__ img, const int32_t* __restrict__ params,
                                                            const float* __restrict__ fparams, float* __restrict__ partial, int Hs, int Ws) {
    __shared__ float red[3][256];
    const int b = blockIdx.y;
    const int32_t* p = params + (int64_t)b * AUG_INTS;
    const int H = p[0], W = p[1], nH = p[2], nW = p[3];
    const float delta = fparams[b * AUG_FLOATS + 0];
    const TI* src = img + (int64_t)b * Hs * Ws * 3;
    float s[3] = {0.f, 0.f, 0.f};
    const int64_t total = (int64_t)nH * nW;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float v[3];
        scaled_pixel(src, Ws, H, W, nH, nW, (int)(i / nW), (int)(i % nW), v);
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] += delta != 0.f ? clip256(v[c] + delta) : v[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) red[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[((int64_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void augment_means_finish_kernel(const float* __restrict__ partial, const int32_t* __restrict__ params, float* __restrict__ fparams,
                                            int nblk) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (c >= 3) return;
    float s = 0.f;
    for (int k = 0; k < nblk; ++k) s += partial[((int64_t)b * nblk + k) * 3 + c];
    const int32_t* p = params + (int64_t)b * AUG_INTS;
    fparams[b * AUG_FLOATS + 2 + c] = s / ((float)p[2] * (float)p[3]);
}

template <class TI>
__global__ __launch_bounds__(256) void augment_crop_kernel(const TI* __restrict__ img, const int32_t* __restrict__ lab,
                                                           const int32_t* __restrict__ params, const float* __restrict__ fparams, AugConst k,
                                                           int ignore, float* __restrict__ out_img, int32_t* __restrict__ out_lab, int B, int Hs,
                                                           int Ws, int ch, int cw, uint64_t seed) {
    const int64_t total = (int64_t)B * ch * cw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % cw), y = (int)((i / cw) % ch), b = (int)(i / ((int64_t)cw * ch));
        const int32_t* p = params + (int64_t)b * AUG_INTS;
        const int H = p[0], W = p[1], nH = p[2], nW = p[3], oy = p[4], ox = p[5], flip = p[6], ne = p[7];
        const int ry = y + oy, rx = (flip ? cw - 1 - x : x) + ox;
        float v[3] = {k.mean[0], k.mean[1], k.mean[2]};
        int l = ignore;
        const float* fp = fparams ? fparams + (int64_t)b * AUG_FLOATS : nullptr;
        if (ry < nH && rx < nW) {
            const TI* src = img + (int64_t)b * Hs * Ws * 3;
            // tf.image.resize(bilinear, half-pixel centres): top + (bottom - top) * ty with top = tl + (tr - tl) * tx
            scaled_pixel(src, Ws, H, W, nH, nW, ry, rx, v);
            if (fp) {      // the photometric augmentations sit between the random scale and the padding (pipeline.py:129-134): image pixels only
                if (fp[0] != 0.f) {      // RandomBrightnessAugment: + delta, clip [0, 256]
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = clip256(v[c] + fp[0]);
                }
                const bool contrast = fp[1] != 1.f, sat_hue = fp[5] != 1.f || fp[6] != 0.f;
                if (contrast) {          // tf.image.adjust_contrast: (x - mean) * factor + mean, mean per channel over the image
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = (v[c] - fp[2 + c]) * fp[1] + fp[2 + c];
                }
                if (sat_hue) saturation_hue(v, fp[5], fp[6]);
                if (contrast || sat_hue) {      // RandomPhotoMetricDistortions ends with its own clip
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = clip256(v[c]);
                }
            }
            if (lab) {      // nearest (v2): min(floor((dst + 0.5) * in / out), in - 1)
                const float fy = (float)H / (float)nH, fx = (float)W / (float)nW;
                const int ly = min((int)floorf(((float)ry + 0.5f) * fy), H - 1), lx = min((int)floorf(((float)rx + 0.5f) * fx), W - 1);
                l = lab[((int64_t)b * Hs + ly) * Ws + lx];
            }
        }
        if (fp && fp[7] > 0.f) {      // RandomNoisyEvalAugment (evaluation pipeline, after the padding): + N(0, sigma), clip [0, 256]
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = clip256(v[c] + fp[7] * aug_normal(seed ^ 0xA0761D6478BD642Full, (uint64_t)i * 3 + c));
        }
        for (int e = 0; e < ne; ++e) {
            const int ey = p[8 + 4 * e], ex = p[9 + 4 * e], eh = p[10 + 4 * e], ew = p[11 + 4 * e];
            if (y >= ey && y < ey + eh && x >= ex && x < ex + ew) {      // a later rectangle overwrites an earlier one, as the loop of the reference
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = 255.f * aug_uniform(seed + (uint64_t)e * 0x632BE59BD9B4E019ull, (uint64_t)i * 3 + c);
                l = ignore;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out_img[i * 3 + c] = v[c] * k.a[c] + k.b[c];
        if (out_lab) out_lab[i] = l;
    }
}

// out = x * a[c] + b[c] on the last axis (3 channels): input normalisation alone (inference)
__global__ void normalize3_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, AugConst k) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        y[i] = x[i] * k.a[c] + k.b[c];
    }
}

}  // namespace

extern "C" int iseg_augment_params_ints(void) { return AUG_INTS; }
extern "C" int iseg_augment_params_floats(void) { return AUG_FLOATS; }

constexpr int AUG_MEAN_BLOCKS = 64;
extern "C" size_t iseg_augment_means_workspace_bytes(int B) { return (size_t)B * AUG_MEAN_BLOCKS * 3 * sizeof(float); }

// fills fparams[b][2..4] with the channel means of every sample's scaled (+ brightness) image: needed before iseg_augment_crop_batch when a
// contrast factor != 1 is in the table
extern "C" int iseg_augment_channel_means(const void* images, int image_dtype, const int32_t* params, float* fparams, int B, int Hs, int Ws,
                                          void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(images && params && fparams && B > 0 && Hs > 0 && Ws > 0, "iseg_augment_channel_means: bad arguments");
    ISEG_REQUIRE(image_dtype == ISEG_F32 || image_dtype == 2, "iseg_augment_channel_means: images must be float32 (0) or uint8 (2)");
    if (!ws || ws_bytes < iseg_augment_means_workspace_bytes(B)) {
        iseg_set_error("iseg_augment_channel_means: needs %zu workspace bytes, got %zu", iseg_augment_means_workspace_bytes(B), ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    if (image_dtype == ISEG_F32)
        hipLaunchKernelGGL((augment_means_kernel<float>), dim3(AUG_MEAN_BLOCKS, B), dim3(256), 0, stream, (const float*)images, params, fparams,
                           (float*)ws, Hs, Ws);
    else
        hipLaunchKernelGGL((augment_means_kernel<uint8_t>), dim3(AUG_MEAN_BLOCKS, B), dim3(256), 0, stream, (const uint8_t*)images, params, fparams,
                           (float*)ws, Hs, Ws);
    hipLaunchKernelGGL(augment_means_finish_kernel, dim3(B), dim3(64), 0, stream, (const float*)ws, params, fparams, AUG_MEAN_BLOCKS);
    return iseg_check_launch("iseg_augment_channel_means");
}

extern "C" int iseg_augment_crop_batch(const void* images, int image_dtype, const int32_t* labels, const int32_t* params, const float* fparams,
                                       const float* mean_pixel, const float* norm_scale, const float* norm_shift, int ignore_label,
                                       float* out_images, int32_t* out_labels, int B, int Hs, int Ws, int crop_h, int crop_w, uint64_t seed,
                                       hipStream_t stream) {
    ISEG_REQUIRE(images && params && mean_pixel && norm_scale && norm_shift && out_images && B > 0 && Hs > 0 && Ws > 0 && crop_h > 0 && crop_w > 0,
                 "iseg_augment_crop_batch: bad arguments");
    ISEG_REQUIRE(image_dtype == ISEG_F32 || image_dtype == 2, "iseg_augment_crop_batch: images must be float32 (0) or uint8 (2)");
    ISEG_REQUIRE((labels != nullptr) == (out_labels != nullptr), "iseg_augment_crop_batch: labels and out_labels go together");
    AugConst k;
    for (int c = 0; c < 3; ++c) k.mean[c] = mean_pixel[c], k.a[c] = norm_scale[c], k.b[c] = norm_shift[c];
    const int64_t total = (int64_t)B * crop_h * crop_w;
    const unsigned blocks = (unsigned)(ceil_div64(total, 256) < 8192 ? ceil_div64(total, 256) : 8192);
    if (image_dtype == ISEG_F32)
        hipLaunchKernelGGL((augment_crop_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)images, labels, params, fparams, k,
                           ignore_label, out_images, out_labels, B, Hs, Ws, crop_h, crop_w, seed);
    else
        hipLaunchKernelGGL((augment_crop_kernel<uint8_t>), dim3(blocks), dim3(256), 0, stream, (const uint8_t*)images, labels, params, fparams, k,
                           ignore_label, out_images, out_labels, B, Hs, Ws, crop_h, crop_w, seed);
    return iseg_check_launch("iseg_augment_crop_batch");
}

extern "C" int iseg_normalize_image(const float* x, float* y, int64_t pixels, const float* norm_scale, const float* norm_shift, hipStream_t stream) {
    ISEG_REQUIRE(x && y && norm_scale && norm_shift && pixels > 0, "iseg_normalize_image: bad arguments");
    AugConst k{};
    for (int c = 0; c < 3; ++c) k.a[c] = norm_scale[c], k.b[c] = norm_shift[c];
    const int64_t n = pixels * 3;
    const unsigned blocks = (unsigned)(ceil_div64(n, 256) < 8192 ? ceil_div64(n, 256) : 8192);
    hipLaunchKernelGGL(normalize3_kernel, dim3(blocks), dim3(256), 0, stream, x, y, n, k);
    return iseg_check_launch("iseg_normalize_image");
}
