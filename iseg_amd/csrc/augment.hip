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

__device__ __forceinline__ float aug_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

template <class TI> __device__ __forceinline__ float px(const TI* p) { return (float)*p; }

struct AugConst {
    float mean[3], a[3], b[3];      // pad colour; out = v * a + b
};

template <class TI>
__global__ __launch_bounds__(256) void augment_crop_kernel(const TI* __restrict__ img, const int32_t* __restrict__ lab,
                                                           const int32_t* __restrict__ params, AugConst k, int ignore, float* __restrict__ out_img,
                                                           int32_t* __restrict__ out_lab, int B, int Hs, int Ws, int ch, int cw, uint64_t seed) {
    const int64_t total = (int64_t)B * ch * cw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % cw), y = (int)((i / cw) % ch), b = (int)(i / ((int64_t)cw * ch));
        const int32_t* p = params + (int64_t)b * AUG_INTS;
        const int H = p[0], W = p[1], nH = p[2], nW = p[3], oy = p[4], ox = p[5], flip = p[6], ne = p[7];
        const int ry = y + oy, rx = (flip ? cw - 1 - x : x) + ox;
        float v[3] = {k.mean[0], k.mean[1], k.mean[2]};
        int l = ignore;
        if (ry < nH && rx < nW) {
            const TI* src = img + (int64_t)b * Hs * Ws * 3;
            // tf.image.resize(bilinear, half-pixel centres): top + (bottom - top) * ty with top = tl + (tr - tl) * tx
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
            if (lab) {      // nearest (v2): min(floor((dst + 0.5) * in / out), in - 1)
                const int ly = min((int)floorf(((float)ry + 0.5f) * fy), H - 1), lx = min((int)floorf(((float)rx + 0.5f) * fx), W - 1);
                l = lab[((int64_t)b * Hs + ly) * Ws + lx];
            }
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

extern "C" int iseg_augment_crop_batch(const void* images, int image_dtype, const int32_t* labels, const int32_t* params, const float* mean_pixel,
                                       const float* norm_scale, const float* norm_shift, int ignore_label, float* out_images,
                                       int32_t* out_labels, int B, int Hs, int Ws, int crop_h, int crop_w, uint64_t seed, hipStream_t stream) {
    ISEG_REQUIRE(images && params && mean_pixel && norm_scale && norm_shift && out_images && B > 0 && Hs > 0 && Ws > 0 && crop_h > 0 && crop_w > 0,
                 "iseg_augment_crop_batch: bad arguments");
    ISEG_REQUIRE(image_dtype == ISEG_F32 || image_dtype == 2, "iseg_augment_crop_batch: images must be float32 (0) or uint8 (2)");
    ISEG_REQUIRE((labels != nullptr) == (out_labels != nullptr), "iseg_augment_crop_batch: labels and out_labels go together");
    AugConst k;
    for (int c = 0; c < 3; ++c) k.mean[c] = mean_pixel[c], k.a[c] = norm_scale[c], k.b[c] = norm_shift[c];
    const int64_t total = (int64_t)B * crop_h * crop_w;
    const unsigned blocks = (unsigned)(ceil_div64(total, 256) < 8192 ? ceil_div64(total, 256) : 8192);
    if (image_dtype == ISEG_F32)
        hipLaunchKernelGGL((augment_crop_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)images, labels, params, k, ignore_label,
                           out_images, out_labels, B, Hs, Ws, crop_h, crop_w, seed);
    else
        hipLaunchKernelGGL((augment_crop_kernel<uint8_t>), dim3(blocks), dim3(256), 0, stream, (const uint8_t*)images, labels, params, k,
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
