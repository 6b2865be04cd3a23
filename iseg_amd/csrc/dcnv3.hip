// DCNv3 core of the reference (layers/dcn_v3/op.py:16-109 + utils.py:14-209), restated exactly, quirks included:
//
//   xp = zero-pad(x, pad) [N, Hin = H + 2 pad, Win = W + 2 pad, G*Cg]          (never materialised: reads are bounds-checked)
//   ref[h]   = ((dil*(k-1))/2 + 0.5 + h*stride) / Hin   and likewise over Win   -> stacked [y, x]            (utils.py:26-56)
//   grid[p]  = (gx[p / kh] / Win, gy[p % kh] / Hin),  gx, gy in {-(dil*(k-1))/2 + i*dil}  -> stacked [x, y]  (utils.py:75-101)
//   e0 = ref_y[h] + grid_x[p]*s + offset[n,h,w,(g*P+p)*2+0] * s / Win                                        (op.py:77-85)
//   e1 = ref_x[w] + grid_y[p]*s + offset[n,h,w,(g*P+p)*2+1] * s / Hin
//   px = 0.5*((2*e0 - 1) + 1)*(Win - 2),  py = 0.5*((2*e1 - 1) + 1)*(Hin - 2)                                 (utils.py:139-143)
//   x0 = floor(px), x1 = x0 + 1, y0 = floor(py), y1 = y0 + 1, each clipped to the padded image; the four bilinear weights are
//   formed from the CLIPPED corners (:146-172), so they can be negative or exceed 1 near the border
//   out[n,h,w,g,:] = sum_p mask[n,h,w,g*P+p] * (wa*xp[y0,x0] + wb*xp[y1,x0] + wc*xp[y0,x1] + wd*xp[y1,x1])[g*Cg:(g+1)*Cg]
//
// i.e. channel 0 of the sampling location mixes the ROW reference with the x offset (for square inputs the base sampling grid is
// transposed) -- reproduced here because parity with the reference is the contract.  Coordinates are computed in fp32 whatever
// the storage dtype (the reference computes them in the compute dtype, a precision hazard under bf16).
//
// One lane = one (n, h, w, g): it walks the P sampling points, 4 corners and Cg channels, so the gradients of offset and mask
// (reductions over the group's channels) need no cross-lane step; the gradient of x is scattered with fp32 atomics.
#include "common.h"
#include "iseg_hip.h"
#include <stdlib.h>
#include <math.h>
#include <type_traits>

namespace {

struct DcnGeom {
    int N, H, W, G, Cg, kh, kw, stride, dil, pad, Hin, Win, Ho, Wo;
    float s;
    // elements between the offset / mask runs of consecutive output pixels (and of their gradients): 2 G P and G P when the two tensors are dense,
    // the row pitch of the joint projection buffer when both live in one [pixels][ld] matrix (iseg_dcnv3_fwd_ld / _bwd_ld)
    int ld_off, ld_mask;
};

struct Tap {
    int x0, x1, y0, y1;          // clipped corners, padded coordinates
    float dx0, dx1, dy0, dy1;    // px - x0, x1 - px, py - y0, y1 - py
};

__device__ __forceinline__ Tap dcn_tap(const DcnGeom& g, int h, int w, int p, float off0, float off1) {
    const float half = (float)((g.dil * (g.kh - 1)) / 2);
    const float halfw = (float)((g.dil * (g.kw - 1)) / 2);
    const float ref_y = (half + 0.5f + (float)(h * g.stride)) / (float)g.Hin;
    const float ref_x = (halfw + 0.5f + (float)(w * g.stride)) / (float)g.Win;
    const float gx = (-halfw + (float)((p / g.kh) * g.dil)) / (float)g.Win;
    const float gy = (-half + (float)((p % g.kh) * g.dil)) / (float)g.Hin;
    const float e0 = ref_y + gx * g.s + off0 * g.s / (float)g.Win;
    const float e1 = ref_x + gy * g.s + off1 * g.s / (float)g.Hin;
    const float px = 0.5f * (((2.f * e0 - 1.f) + 1.0f) * (float)(g.Win - 2));
    const float py = 0.5f * (((2.f * e1 - 1.f) + 1.0f) * (float)(g.Hin - 2));
    Tap t;
    const int fx = (int)floorf(px), fy = (int)floorf(py);
    t.x0 = min(max(fx, 0), g.Win - 1);
    t.x1 = min(max(fx + 1, 0), g.Win - 1);
    t.y0 = min(max(fy, 0), g.Hin - 1);
    t.y1 = min(max(fy + 1, 0), g.Hin - 1);
    t.dx0 = px - (float)t.x0;
    t.dx1 = (float)t.x1 - px;
    t.dy0 = py - (float)t.y0;
    t.dy1 = (float)t.y1 - py;
    return t;
}

// element offset of padded pixel (y, x) in the UNPADDED tensor, or -1 inside the zero ring
__device__ __forceinline__ int64_t dcn_src(const DcnGeom& g, int n, int y, int x) {
    const int uy = y - g.pad, ux = x - g.pad;
    if ((unsigned)uy >= (unsigned)g.H || (unsigned)ux >= (unsigned)g.W) return -1;
    return (((int64_t)n * g.H + uy) * g.W + ux) * (g.G * g.Cg);
}

template <class T, int CV>
__device__ __forceinline__ void ldv(const T* p, float* v) {
    if (CV == 8) load8<T>(p, v);
    else v[0] = to_f32(p[0]);
}

constexpr int DCN_PMAX = 9;                       // sampling points the pipelined kernels keep in registers (3 x 3)

__device__ __forceinline__ void unpack_raw_bf16x8(const uint4& r, float* v) {
    v[0] = __uint_as_float(r.x << 16), v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16), v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16), v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16), v[7] = __uint_as_float(r.w & 0xffff0000u);
}

template <class T, int CG> __device__ __forceinline__ void dcn_unpack_row(const uint4* raw, float* v) {
    constexpr int RAW = CG * (int)sizeof(T) / 16;
#pragma unroll
    for (int r = 0; r < RAW; ++r) {
        if constexpr (sizeof(T) == 2) unpack_raw_bf16x8(raw[r], v + r * 8);
        else {
            v[r * 4] = __uint_as_float(raw[r].x), v[r * 4 + 1] = __uint_as_float(raw[r].y);
            v[r * 4 + 2] = __uint_as_float(raw[r].z), v[r * 4 + 3] = __uint_as_float(raw[r].w);
        }
    }
}

// Forward for the group widths of the reference's models (Cg 8 / 16, at most 3 x 3 points): a lane owns (pixel, group) and all Cg channels.
// The kernel is a chain of dependent round trips (offsets -> tap -> four corner rows, per point), so the offsets and masks of all points are
// loaded up front and the corner rows of point p + 1 are requested before point p is accumulated; loads are issued unconditionally from a
// clamped address and masked afterwards.  8 x 128 x 128 x 112: 457 -> see DESIGN 5.1.
template <class T, int CG>
__global__ __launch_bounds__(256) void dcnv3_fwd_pipe_kernel(const T* __restrict__ x, const T* __restrict__ offset, const T* __restrict__ mask,
                                                             T* __restrict__ y, DcnGeom g) {
    constexpr int RAW = CG * (int)sizeof(T) / 16;
    const int P = g.kh * g.kw;
    const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.G;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int gi = (int)(i % g.G);
        int64_t t = i / g.G;
        const int w = (int)(t % g.Wo);
        t /= g.Wo;
        const int h = (int)(t % g.Ho);
        const int n = (int)(t / g.Ho);
        float offs[DCN_PMAX][2], mk[DCN_PMAX];
        const T* const op = offset + (i / g.G) * g.ld_off + gi * P * 2;
        const T* const mp = mask + (i / g.G) * g.ld_mask + gi * P;
#pragma unroll
        for (int p = 0; p < DCN_PMAX; ++p) {
            const int pp = p < P ? p : P - 1;
            offs[p][0] = to_f32(op[pp * 2]);
            offs[p][1] = to_f32(op[pp * 2 + 1]);
            mk[p] = to_f32(mp[pp]);
        }
        uint4 nxt[4][RAW];
        Tap ntp{};
        bool nok[4] = {false, false, false, false};
        auto request = [&](int p) {
            ntp = dcn_tap(g, h, w, p, offs[p][0], offs[p][1]);
            const int ys[4] = {ntp.y0, ntp.y1, ntp.y0, ntp.y1};
            const int xs[4] = {ntp.x0, ntp.x0, ntp.x1, ntp.x1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t src = dcn_src(g, n, ys[k], xs[k]);
                nok[k] = src >= 0;
                const uint4* q = reinterpret_cast<const uint4*>(x + (src < 0 ? 0 : src) + gi * CG);
#pragma unroll
                for (int r = 0; r < RAW; ++r) nxt[k][r] = q[r];
            }
        };
        request(0);
        float acc[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[c] = 0.f;
#pragma unroll
        for (int p = 0; p < DCN_PMAX; ++p) {
            if (p >= P) continue;
            const Tap tp = ntp;
            float v[4][CG];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ok[k] = nok[k];
                dcn_unpack_row<T, CG>(nxt[k], v[k]);
            }
            if (p + 1 < P) request(p + 1);
            const float m = mk[p];
            const float wgt[4] = {ok[0] ? m * tp.dx1 * tp.dy1 : 0.f, ok[1] ? m * tp.dx1 * tp.dy0 : 0.f, ok[2] ? m * tp.dx0 * tp.dy1 : 0.f,
                                  ok[3] ? m * tp.dx0 * tp.dy0 : 0.f};
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                // (the reference sums the four corners first and applies the mask to the sum; the product order differs in the last bit only)
                float px = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) px = fmaf(wgt[k], v[k][c], px);
                acc[c] += px;
            }
        }
        T* yp = y + i * CG;
#pragma unroll
        for (int c0 = 0; c0 < CG; c0 += 8) store8<T>(yp + c0, acc + c0);
    }
}

template <class T, int CV>
__global__ __launch_bounds__(256) void dcnv3_fwd_kernel(const T* __restrict__ x, const T* __restrict__ offset, const T* __restrict__ mask,
                                                        T* __restrict__ y, DcnGeom g) {
    const int P = g.kh * g.kw;
    const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.G;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int gi = (int)(i % g.G);
        int64_t t = i / g.G;
        const int w = (int)(t % g.Wo);
        t /= g.Wo;
        const int h = (int)(t % g.Ho);
        const int n = (int)(t / g.Ho);
        const int64_t pix = ((int64_t)n * g.Ho + h) * g.Wo + w;
        const T* op = offset + pix * g.ld_off + gi * P * 2;
        const T* mp = mask + pix * g.ld_mask + gi * P;
        T* yp = y + (pix * g.G + gi) * g.Cg;
        for (int c0 = 0; c0 < g.Cg; c0 += CV) {
            float acc[CV];
#pragma unroll
            for (int u = 0; u < CV; ++u) acc[u] = 0.f;
            for (int p = 0; p < P; ++p) {
                const Tap tp = dcn_tap(g, h, w, p, to_f32(op[2 * p]), to_f32(op[2 * p + 1]));
                const float m = to_f32(mp[p]);
                const float wgt[4] = {tp.dx1 * tp.dy1, tp.dx1 * tp.dy0, tp.dx0 * tp.dy1, tp.dx0 * tp.dy0};
                const int ys[4] = {tp.y0, tp.y1, tp.y0, tp.y1};
                const int xs[4] = {tp.x0, tp.x0, tp.x1, tp.x1};
                float px[CV];
#pragma unroll
                for (int u = 0; u < CV; ++u) px[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t src = dcn_src(g, n, ys[k], xs[k]);
                    if (src >= 0) {
                        float v[CV];
                        ldv<T, CV>(x + src + gi * g.Cg + c0, v);
#pragma unroll
                        for (int u = 0; u < CV; ++u) px[u] = fmaf(wgt[k], v[u], px[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < CV; ++u) acc[u] = fmaf(px[u], m, acc[u]);
            }
            if (CV == 8) store8<T>(yp + c0, acc);
            else yp[c0] = from_f32<T>(acc[0]);
        }
    }
}

// 2^-40 fixed point (int64) for every input-gradient accumulator of this file: integer adds commute, so atomics give the same bits whatever
// the order (the reference's default is a deterministic step, core_env.py:39-48).  |v| < 2^23, resolution 9.1e-13.
constexpr float DCN_FIX = 256.f;                  // 2^8: high word = floor(v * 2^8), low word = fract * 2^32
constexpr double DCN_UNFIX = 1.0 / 1099511627776.0;      // 2^-40
// Sign and magnitude: the magnitude's integer part and fraction are both exact in fp32 (a - floor(a) of a non-negative a needs no more bits than
// a has), so the only error is the rounding at 2^-40.  (Rounds 4-5 split the SIGNED value: floor(v) = -1 for a small negative v and the
// fraction v + 1 was rounded to fp32's 2^-24 next to 1 -- a resolution of 2^-32 = 2.3e-10 instead of 9.1e-13 for every negative contribution,
// which the 512 x 512 parity of round 6 found: gradient elements of 1e-9 under a mean loss over 262 144 pixels were off by per cent.)
__device__ __forceinline__ unsigned long long dcn_to_fixed(float v256) {      // v256 = value * 2^8
    const float a = fabsf(v256);
    const float fl = floorf(a);
    const unsigned hi = (unsigned)(int)fl;                      // (int): saturating
    const unsigned lo = (unsigned)rintf((a - fl) * 4294967296.f);      // nearest (truncation shrinks every contribution: a bias; 2^32 saturates one quantum low)
    const unsigned long long mag = ((unsigned long long)hi << 32) | lo;
    return v256 < 0.f ? 0ull - mag : mag;
}

// gradients: dx accumulated as int64 fixed point by global integer atomics into a zeroed workspace (order-free), converted to fp32 by
// dcn_unfix_kernel; doffset, dmask written once per (n,h,w,g,p)
template <class T, int CV>
__global__ __launch_bounds__(256) void dcnv3_bwd_kernel(const T* __restrict__ x, const T* __restrict__ offset, const T* __restrict__ mask,
                                                        const T* __restrict__ dy, unsigned long long* __restrict__ dx, T* __restrict__ doffset,
                                                        T* __restrict__ dmask, int* __restrict__ flag, DcnGeom g) {
    const int P = g.kh * g.kw;
    const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.G;
    bool bad = false;      // a non-finite contribution: the fixed-point conversion would saturate it to finite garbage (see dcn_unfix_kernel)
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int gi = (int)(i % g.G);
        int64_t t = i / g.G;
        const int w = (int)(t % g.Wo);
        t /= g.Wo;
        const int h = (int)(t % g.Ho);
        const int n = (int)(t / g.Ho);
        const int64_t pix = ((int64_t)n * g.Ho + h) * g.Wo + w;
        const T* op = offset + pix * g.ld_off + gi * P * 2;
        const T* mp = mask + pix * g.ld_mask + gi * P;
        const T* dyp = dy + (pix * g.G + gi) * g.Cg;
        for (int p = 0; p < P; ++p) {
            const Tap tp = dcn_tap(g, h, w, p, to_f32(op[2 * p]), to_f32(op[2 * p + 1]));
            const float m = to_f32(mp[p]);
            const float wgt[4] = {tp.dx1 * tp.dy1, tp.dx1 * tp.dy0, tp.dx0 * tp.dy1, tp.dx0 * tp.dy0};
            // d weight / d px and / d py (floor and clip carry no gradient)
            const float wpx[4] = {-tp.dy1, -tp.dy0, tp.dy1, tp.dy0};
            const float wpy[4] = {-tp.dx1, tp.dx1, -tp.dx0, tp.dx0};
            const int ys[4] = {tp.y0, tp.y1, tp.y0, tp.y1};
            const int xs[4] = {tp.x0, tp.x0, tp.x1, tp.x1};
            float gm = 0.f, gpx = 0.f, gpy = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t src = dcn_src(g, n, ys[k], xs[k]);
                if (src < 0) continue;
                float dot = 0.f;    // sum_c dy[c] * xp[corner][c]
                for (int c0 = 0; c0 < g.Cg; c0 += CV) {
                    float v[CV], d[CV];
                    ldv<T, CV>(x + src + gi * g.Cg + c0, v);
                    ldv<T, CV>(dyp + c0, d);
#pragma unroll
                    for (int u = 0; u < CV; ++u) {
                        dot = fmaf(d[u], v[u], dot);
                        const float contrib = d[u] * m * wgt[k] * DCN_FIX;
                        bad |= !(fabsf(contrib) < 3.0e38f);
                        atomicAdd(dx + src + gi * g.Cg + c0 + u, dcn_to_fixed(contrib));
                    }
                }
                gm = fmaf(wgt[k], dot, gm);
                gpx = fmaf(wpx[k], dot, gpx);
                gpy = fmaf(wpy[k], dot, gpy);
            }
            dmask[pix * g.ld_mask + gi * P + p] = from_f32<T>(gm);
            // px = e0 * (Win - 2), e0 = ... + off0 * s / Win
            doffset[pix * g.ld_off + (gi * P + p) * 2] = from_f32<T>(gpx * m * (float)(g.Win - 2) * g.s / (float)g.Win);
            doffset[pix * g.ld_off + (gi * P + p) * 2 + 1] = from_f32<T>(gpy * m * (float)(g.Hin - 2) * g.s / (float)g.Hin);
        }
    }
    if (bad) atomicOr(flag, 2);      // (rare; integer OR: order-free)
}

// Channel-lane variant of the backward pass for power-of-two group widths (InternImage: Cg = 16 everywhere): LC = Cg lanes
// share one (pixel, group) -- lane c owns channel c -- so every load and every atomic of a corner is one contiguous Cg-element
// run (16 x fewer L2 atomic requests than one lane per (pixel, group): 9.8 ms -> see profiles), and the channel reductions for
// dmask / doffset are log2(LC) shuffles.  The tap geometry is recomputed by each of the LC lanes (cheap ALU).
template <class T, int LC>
__global__ __launch_bounds__(256) void dcnv3_bwd_cl_kernel(const T* __restrict__ x, const T* __restrict__ offset,
                                                           const T* __restrict__ mask, const T* __restrict__ dy,
                                                           unsigned long long* __restrict__ dx, T* __restrict__ doffset,
                                                           T* __restrict__ dmask, int* __restrict__ flag, DcnGeom g) {
    const int P = g.kh * g.kw;
    const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.G;          // (pixel, group) items
    const int c = threadIdx.x % LC;
    bool bad = false;      // (as in dcnv3_bwd_kernel)
    constexpr int IPB = 256 / LC;                                     // items per workgroup sweep
    for (int64_t i = blockIdx.x * (int64_t)IPB + threadIdx.x / LC; i < total; i += (int64_t)gridDim.x * IPB) {
        const int gi = (int)(i % g.G);
        int64_t t = i / g.G;
        const int w = (int)(t % g.Wo);
        t /= g.Wo;
        const int h = (int)(t % g.Ho);
        const int n = (int)(t / g.Ho);
        const int64_t pix = ((int64_t)n * g.Ho + h) * g.Wo + w;
        const T* op = offset + pix * g.ld_off + gi * P * 2;
        const T* mp = mask + pix * g.ld_mask + gi * P;
        const float d = to_f32(dy[(pix * g.G + gi) * g.Cg + c]);
        for (int p = 0; p < P; ++p) {
            const Tap tp = dcn_tap(g, h, w, p, to_f32(op[2 * p]), to_f32(op[2 * p + 1]));
            const float m = to_f32(mp[p]);
            const float wgt[4] = {tp.dx1 * tp.dy1, tp.dx1 * tp.dy0, tp.dx0 * tp.dy1, tp.dx0 * tp.dy0};
            const float wpx[4] = {-tp.dy1, -tp.dy0, tp.dy1, tp.dy0};
            const float wpy[4] = {-tp.dx1, tp.dx1, -tp.dx0, tp.dx0};
            const int ys[4] = {tp.y0, tp.y1, tp.y0, tp.y1};
            const int xs[4] = {tp.x0, tp.x0, tp.x1, tp.x1};
            float gm = 0.f, gpx = 0.f, gpy = 0.f;
            const float dm = d * m;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t src = dcn_src(g, n, ys[k], xs[k]);
                if (src < 0) continue;        // uniform across the LC lanes of an item
                const float v = to_f32(x[src + gi * g.Cg + c]);
                const float dv = d * v;       // this lane's share of sum_c dy[c] * xp[corner][c]
                const float contrib = dm * wgt[k] * DCN_FIX;
                bad |= !(fabsf(contrib) < 3.0e38f);
                atomicAdd(dx + src + gi * g.Cg + c, dcn_to_fixed(contrib));
                gm = fmaf(wgt[k], dv, gm);
                gpx = fmaf(wpx[k], dv, gpx);
                gpy = fmaf(wpy[k], dv, gpy);
            }
            gm = group_sum(gm, LC);
            gpx = group_sum(gpx, LC);
            gpy = group_sum(gpy, LC);
            if (c == 0) {
                dmask[pix * g.ld_mask + gi * P + p] = from_f32<T>(gm);
                doffset[pix * g.ld_off + (gi * P + p) * 2] = from_f32<T>(gpx * m * (float)(g.Win - 2) * g.s / (float)g.Win);
                doffset[pix * g.ld_off + (gi * P + p) * 2 + 1] = from_f32<T>(gpy * m * (float)(g.Hin - 2) * g.s / (float)g.Hin);
            }
        }
    }
    if (bad) atomicOr(flag, 2);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Deterministic backward pass (Cg in {8, 16}): no floating-point atomics anywhere.
//   * a workgroup owns (image, 16 x 16 output tile, group) and an LDS window of WS x WS padded-input pixels around the tile's zero-offset
//     sampling footprint (the footprint of output rows lies along the input's x axis and vice versa -- the reference's transposed base
//     grid -- so the window's x origin follows the tile's row index);
//   * lane = (output pixel, sampling point p): tap geometry once, the group's Cg channels in registers -- the offset / mask gradients
//     need no cross-lane step;
//   * the input gradient is accumulated in 2^-40 fixed point (int64): integer adds commute, so the LDS atomics (ds_add_u64) give the
//     same bits whatever the order.  Samples that leave the window (|offset| beyond the window margin) go to a dense int64 side
//     buffer with global integer atomics -- equally order-free, only slower;
//   * the window is written out as fp32 and a second kernel sums, for every input pixel, the windows covering it in tile order and adds
//     the side buffer when any workgroup used it.
// Values are clamped to |v| < 2^23 by the fixed-point conversion (resolution 9.1e-13).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int DCN_TS = 16, DCN_WS = 24;
// LDS image of a window: pixel (ly, lx), channel c at accumulator index ly * RP + lx * PP + c with PP = CG + 1 and RP = WS * PP + 1 (odd
// pitches: the lanes of one LDS atomic -- a fixed channel, 16 x 4 neighbouring output pixels whose footprints step along ly for consecutive
// lanes and along lx for consecutive rows -- spread over the banks, and the channel is an IMMEDIATE offset of the instruction.  The rotated
// slot (c + ly + lx) mod CG of the first form cost three VALU instructions of address arithmetic per atomic: 1 700 of a wavefront's 7 700.)
template <int CG> struct DcnWinLds {
    static constexpr int PP = CG + 1, RP = DCN_WS * PP + 1, CELLS = DCN_WS * RP;
};

struct DcnWin {
    int tiles_y, tiles_x;      // output tiles
    int c0x, c0y, rx, ry;      // window origin: x0(ty) = ty*TS*stride*(Win-2)/Hin + c0x - rx,  y0(tx) = tx*TS*stride*(Hin-2)/Win + c0y - ry
};

__device__ __forceinline__ int dcn_win_x0(const DcnGeom& g, const DcnWin& wn, int ty) {
    return (ty * DCN_TS * g.stride * (g.Win - 2)) / g.Hin + wn.c0x - wn.rx;
}
__device__ __forceinline__ int dcn_win_y0(const DcnGeom& g, const DcnWin& wn, int tx) {
    return (tx * DCN_TS * g.stride * (g.Hin - 2)) / g.Win + wn.c0y - wn.ry;
}

// bf16 storage (FIX32): the window accumulates 32-bit integers on a per-workgroup power-of-two scale instead -- half the LDS (four workgroups
// per CU), ds_add_u32 instead of ds_add_u64, a multiply + round per addend instead of the two-word split.  A cell receives at most
// 256 pixels x 9 points x 4 corners < 2^14 addends, each bounded by B = max|dy| * max|mask| over the workgroup's lanes (bilinear weights
// <= 1), so with scale = 2^(30 - e), B * 2^14 < 2^e, no sum can leave int32; every addend is rounded to 2^-16 of B or finer -- two orders
// below the bf16 rounding of the result -- and integer adds commute, so the result is still independent of the order.
template <class T, int CG>
__global__ __launch_bounds__(256) void dcnv3_bwd_win_kernel(const T* __restrict__ x, const T* __restrict__ offset, const T* __restrict__ mask,
                                                            const T* __restrict__ dy, T* __restrict__ doffset, T* __restrict__ dmask,
                                                            float* __restrict__ windows, unsigned long long* __restrict__ side,
                                                            int* __restrict__ side_used, DcnGeom g, DcnWin wn) {
    constexpr bool FIX32 = sizeof(T) == 2;
    using Acc = typename std::conditional<FIX32, int, unsigned long long>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned long long win_raw[];
    // window image: DcnWinLds (padded pitches).
    // (Kept unrolled over the points at 161 registers / three workgroups per CU: capped at 128 for a fourth it spills 54-66 registers and
    // runs 1.5x slower, rolled or not.)
    using WL = DcnWinLds<CG>;
    Acc* const win_acc = reinterpret_cast<Acc*>(win_raw);
    T* const stage = reinterpret_cast<T*>(win_acc + WL::CELLS);      // [256 pixels][P mask gradients | 2 P offset gradients], behind the window
    __shared__ float wave_max[4];
    constexpr int WPIX = DCN_WS * DCN_WS;
    for (int i = threadIdx.x; i < WL::CELLS; i += 256) win_acc[i] = (Acc)0;
    int b = blockIdx.x;
    const int gi = b % g.G;
    b /= g.G;
    const int tx = b % wn.tiles_x;
    b /= wn.tiles_x;
    const int ty = b % wn.tiles_y;
    const int n = b / wn.tiles_y;
    const int wx0 = dcn_win_x0(g, wn, ty), wy0 = dcn_win_y0(g, wn, tx);
    const int h = ty * DCN_TS + (threadIdx.x >> 4), w = tx * DCN_TS + (threadIdx.x & 15);
    const bool live = h < g.Ho && w < g.Wo;
    const int P = g.kh * g.kw;
    const int64_t pix = ((int64_t)n * g.Ho + h) * g.Wo + w;
    float d[CG];
    constexpr int RAWD = CG * (int)sizeof(T) / 16;
    uint4 draw[RAWD];      // the same row as stored (bf16 pairs: operands of v_dot2_f32_bf16)
#pragma unroll
    for (int r = 0; r < RAWD; ++r) draw[r] = make_uint4(0u, 0u, 0u, 0u);
    if (live) {
#pragma unroll
        for (int r = 0; r < RAWD; ++r) draw[r] = reinterpret_cast<const uint4*>(dy + (pix * g.G + gi) * CG)[r];
    }
    dcn_unpack_row<T, CG>(draw, d);
    bool spilled = false;
    // every sampling point's offsets and mask up front (one round trip instead of one per point; P <= DCN_PMAX is a condition of this path)
    float offs[DCN_PMAX][2], mk[DCN_PMAX];
#pragma unroll
    for (int p = 0; p < DCN_PMAX; ++p) {
        const int pp = p < P ? p : P - 1;
        const int64_t eo = live ? pix * g.ld_off + (gi * P + pp) * 2 : 0, em = live ? pix * g.ld_mask + gi * P + pp : 0;
        offs[p][0] = to_f32(offset[eo]);
        offs[p][1] = to_f32(offset[eo + 1]);
        mk[p] = to_f32(mask[em]);
    }
    float scale = 0.f, inv_scale = 0.f;
    {
        // a non-finite arriving gradient or mask must stay visible: the integer accumulators would turn it into finite garbage, so the workgroup
        // raises bit 1 of the flag word and the gather kernel then writes NaN for the whole input gradient (what fp32 atomics used to spread)
        float s = 0.f;
        if (live) {
#pragma unroll
            for (int c = 0; c < CG; ++c) s += d[c] * 0.f;      // 0 for finite values, NaN for inf / NaN
#pragma unroll
            for (int p = 0; p < DCN_PMAX; ++p) s += mk[p] * 0.f;
        }
        if (s != s) atomicOr(side_used, 2);
    }
    if constexpr (FIX32) {
        float bd = 0.f, bm = 0.f;
        if (live) {
#pragma unroll
            for (int c = 0; c < CG; ++c) bd = fmaxf(bd, fabsf(d[c]));
#pragma unroll
            for (int p = 0; p < DCN_PMAX; ++p) bm = fmaxf(bm, fabsf(mk[p]));
        }
        float bound = bd * bm;
        if (!(bound < 3.0e38f)) bound = 3.0e38f;      // (inf / NaN gradients: the sums saturate instead of wrapping; the flag above reports them)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o));
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = bound;
        __syncthreads();
        bound = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        int e = 0;
        (void)frexpf(bound, &e);                      // bound < 2^e
        int k = 30 - 14 - e;
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
        scale = bound > 0.f ? ldexpf(1.f, k) : 0.f;
        inv_scale = bound > 0.f ? ldexpf(1.f, -k) : 0.f;
    }
    __syncthreads();
    // the four corner rows of point p + 1 are requested before point p's dot products and atomics: loads are issued unconditionally from a
    // clamped address (a branch per load would make the compiler wait for each) and masked afterwards
    constexpr int RAW = CG * (int)sizeof(T) / 16;
    uint4 nxt[4][RAW];
    Tap ntp{};
    int64_t nsrc[4] = {-1, -1, -1, -1};
    auto request = [&](int p) {
        ntp = dcn_tap(g, h, w, p, offs[p][0], offs[p][1]);
        const int ys[4] = {ntp.y0, ntp.y1, ntp.y0, ntp.y1};
        const int xs[4] = {ntp.x0, ntp.x0, ntp.x1, ntp.x1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            nsrc[k] = dcn_src(g, n, ys[k], xs[k]);
            const uint4* q = reinterpret_cast<const uint4*>(x + (nsrc[k] < 0 ? 0 : nsrc[k]) + gi * CG);
#pragma unroll
            for (int r = 0; r < RAW; ++r) nxt[k][r] = q[r];
        }
    };
    if (live) request(0);
#pragma unroll
    for (int p = 0; p < DCN_PMAX; ++p) {
        if (p >= P || !live) continue;      // (P is uniform; dead lanes belong to partial edge tiles)
        const Tap tp = ntp;
        int64_t srcs[4];
        float dots[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            srcs[k] = nsrc[k];
            // <dy, x corner> over the group's channels: bf16 pairs straight into v_dot2_f32_bf16 (exact products, fp32 sums) -- no unpack
            float dot = 0.f;
            if constexpr (FIX32) {
#pragma unroll
                for (int r = 0; r < RAW; ++r) {
                    dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, draw[r].x), __builtin_bit_cast(bf16x2_t, nxt[k][r].x), dot, false);
                    dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, draw[r].y), __builtin_bit_cast(bf16x2_t, nxt[k][r].y), dot, false);
                    dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, draw[r].z), __builtin_bit_cast(bf16x2_t, nxt[k][r].z), dot, false);
                    dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, draw[r].w), __builtin_bit_cast(bf16x2_t, nxt[k][r].w), dot, false);
                }
            } else {
                float v[CG];
                dcn_unpack_row<T, CG>(nxt[k], v);
#pragma unroll
                for (int c = 0; c < CG; ++c) dot = fmaf(d[c], v[c], dot);
            }
            dots[k] = dot;
        }
        if (p + 1 < P) request(p + 1);
        const float m = mk[p];
        const float wgt[4] = {tp.dx1 * tp.dy1, tp.dx1 * tp.dy0, tp.dx0 * tp.dy1, tp.dx0 * tp.dy0};
        const float wpx[4] = {-tp.dy1, -tp.dy0, tp.dy1, tp.dy0};
        const float wpy[4] = {-tp.dx1, tp.dx1, -tp.dx0, tp.dx0};
        const int ys[4] = {tp.y0, tp.y1, tp.y0, tp.y1};
        const int xs[4] = {tp.x0, tp.x0, tp.x1, tp.x1};
        float gm = 0.f, gpx = 0.f, gpy = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (srcs[k] < 0) continue;
            const float dot = dots[k];
            gm = fmaf(wgt[k], dot, gm);
            gpx = fmaf(wpx[k], dot, gpx);
            gpy = fmaf(wpy[k], dot, gpy);
            const int lx = xs[k] - wx0, ly = ys[k] - wy0;
            if ((unsigned)lx < (unsigned)DCN_WS && (unsigned)ly < (unsigned)DCN_WS) {
                Acc* dst = win_acc + ly * WL::RP + lx * WL::PP;      // (+ c: an immediate of the ds_add)
                if constexpr (FIX32) {
                    const float c32 = m * wgt[k] * scale;
#pragma unroll
                    for (int c = 0; c < CG; ++c)      // round to nearest through the 1.5 * 2^23 binade: |addend| < 2^16 by the scale
                        atomicAdd(dst + c, __float_as_int(fmaf(d[c], c32, 12582912.f)) - 0x4B400000);
                } else {
                    const float coef = m * wgt[k] * DCN_FIX;
#pragma unroll
                    for (int c = 0; c < CG; ++c) atomicAdd(dst + c, dcn_to_fixed(d[c] * coef));
                }
            } else {
                spilled = true;
                const float coef = m * wgt[k] * DCN_FIX;
                unsigned long long* dst = side + srcs[k] + gi * CG;
#pragma unroll
                for (int c = 0; c < CG; ++c) atomicAdd(dst + c, dcn_to_fixed(d[c] * coef));
            }
        }
        // staged in LDS and written after the loop: a lane's own stores (one element of a pixel's run per point, pixels 2 G P bytes apart)
        // were 64 partial cache lines per instruction -- 194 of the kernel's 540 us at 8 x 128 x 128 x 112, 3.5x its algorithmic write traffic
        T* const st = stage + threadIdx.x * (3 * DCN_PMAX);
        st[p] = from_f32<T>(gm);
        st[DCN_PMAX + 2 * p] = from_f32<T>(gpx * m * (float)(g.Win - 2) * g.s / (float)g.Win);
        st[DCN_PMAX + 2 * p + 1] = from_f32<T>(gpy * m * (float)(g.Hin - 2) * g.s / (float)g.Hin);
    }
    if (spilled) atomicOr(side_used, 1);
    __syncthreads();
    // the staged mask / offset gradients: consecutive lanes write consecutive elements of a pixel's run (P mask values, 2 P offset values)
    for (int e = threadIdx.x; e < 256 * P; e += 256) {
        const int pl = e / P, p = e - pl * P;
        const int hh = ty * DCN_TS + (pl >> 4), ww = tx * DCN_TS + (pl & 15);
        if (hh < g.Ho && ww < g.Wo) {
            const int64_t opix = ((int64_t)n * g.Ho + hh) * g.Wo + ww;
            const int64_t run_m = opix * g.ld_mask + gi * P + p, run_o = opix * g.ld_off + (gi * P + p) * 2;      // (ld_off even: checked on the host)
            const T* sp = stage + pl * (3 * DCN_PMAX);
            dmask[run_m] = sp[p];
            if constexpr (sizeof(T) == 2) {      // the (x, y) pair of a point as one dword
                const unsigned pair = (unsigned)__builtin_bit_cast(unsigned short, sp[DCN_PMAX + 2 * p]) |
                                      ((unsigned)__builtin_bit_cast(unsigned short, sp[DCN_PMAX + 2 * p + 1]) << 16);
                *reinterpret_cast<unsigned*>(doffset + run_o) = pair;
            } else {
                doffset[run_o] = sp[DCN_PMAX + 2 * p];
                doffset[run_o + 1] = sp[DCN_PMAX + 2 * p + 1];
            }
        }
    }
    float* out = windows + (int64_t)blockIdx.x * WPIX * CG;
    for (int i = threadIdx.x; i < WPIX * CG; i += 256) {
        const int wp = i / CG, c = i % CG;
        const int cell = (wp / DCN_WS) * WL::RP + (wp % DCN_WS) * WL::PP + c;
        if constexpr (FIX32) out[i] = (float)win_acc[cell] * inv_scale;
        else out[i] = (float)((double)(long long)win_acc[cell] * DCN_UNFIX);
    }
}

// dx[n, uy, ux, g, :] = sum of the windows that cover padded pixel (uy + pad, ux + pad), tile-row-major, + the side buffer
// TO: the storage type of dx (fp32, or bf16 directly: no fp32 image and cast pass in between).  RESET: the side buffer is a caller-kept one that is
// all zero between calls (no 8-B-per-element zeroing pass per backward): a lane that consumed its entries writes the zeros back.
template <int CG, class TO, bool RESET>
__global__ __launch_bounds__(256) void dcnv3_bwd_gather_kernel(const float* __restrict__ windows, unsigned long long* __restrict__ side,
                                                               const int* __restrict__ side_used, TO* __restrict__ dx, DcnGeom g,
                                                               DcnWin wn) {
    constexpr int WPIX = DCN_WS * DCN_WS, Q = CG / 4;
    // (the host keeps N H W G Q and the bracket products below 2^31: 32-bit index arithmetic -- the 64-bit divisions of the first form were
    // most of this kernel's 600 VALU instructions per wavefront)
    const unsigned total = (unsigned)g.N * g.H * g.W * g.G * Q;
    const int flags = *side_used;      // bit 0: some workgroup spilled into the side buffer; bit 1: a non-finite gradient / mask was seen
    const bool use_side = (flags & 1) != 0, poisoned = (flags & 2) != 0;
    const int kx = DCN_TS * g.stride * (g.Win - 2), ky = DCN_TS * g.stride * (g.Hin - 2);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int qc = (int)(i % (unsigned)Q);
        unsigned t = i / (unsigned)Q;
        const int gi = (int)(t % (unsigned)g.G);
        t /= (unsigned)g.G;
        const int ux = (int)(t % (unsigned)g.W);
        t /= (unsigned)g.W;
        const int uy = (int)(t % (unsigned)g.H);
        const int n = (int)(t / (unsigned)g.H);
        const int px = ux + g.pad, py = uy + g.pad;
        // tiles whose window may hold (py, px): x0(ty) = floor(ty * kx / Hin) + c0x - rx grows with the tile ROW index; bracket the solutions
        // of x0(ty) <= px < x0(ty) + WS with one tile of slack and test exactly below
        const int ax = px - wn.c0x + wn.rx, ay = py - wn.c0y + wn.ry;
        const int ty_hi = min(wn.tiles_y - 1, kx > 0 ? ((ax + 1) * g.Hin) / kx + 1 : wn.tiles_y - 1);
        const int ty_lo = max(0, kx > 0 ? ((ax - DCN_WS) * g.Hin) / kx - 1 : 0);
        const int tx_hi = min(wn.tiles_x - 1, ky > 0 ? ((ay + 1) * g.Win) / ky + 1 : wn.tiles_x - 1);
        const int tx_lo = max(0, ky > 0 ? ((ay - DCN_WS) * g.Win) / ky - 1 : 0);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ty = ty_lo; ty <= ty_hi; ++ty) {
            const int lx = px - dcn_win_x0(g, wn, ty);
            if ((unsigned)lx >= (unsigned)DCN_WS) continue;
            for (int tx = tx_lo; tx <= tx_hi; ++tx) {
                const int ly = py - dcn_win_y0(g, wn, tx);
                if ((unsigned)ly >= (unsigned)DCN_WS) continue;
                const int64_t blk = (((int64_t)n * wn.tiles_y + ty) * wn.tiles_x + tx) * g.G + gi;
                const float4 v = *reinterpret_cast<const float4*>(windows + (blk * WPIX + ly * DCN_WS + lx) * CG + qc * 4);
                acc.x += v.x;
                acc.y += v.y;
                acc.z += v.z;
                acc.w += v.w;
            }
        }
        const int64_t e = ((((int64_t)n * g.H + uy) * g.W + ux) * g.G + gi) * CG + qc * 4;
        if (use_side) {
            acc.x += (float)((double)(long long)side[e] * DCN_UNFIX);
            acc.y += (float)((double)(long long)side[e + 1] * DCN_UNFIX);
            acc.z += (float)((double)(long long)side[e + 2] * DCN_UNFIX);
            acc.w += (float)((double)(long long)side[e + 3] * DCN_UNFIX);
            if (RESET) side[e] = side[e + 1] = side[e + 2] = side[e + 3] = 0ull;
        }
        if (poisoned) acc.x = acc.y = acc.z = acc.w = __builtin_nanf("");
        if constexpr (sizeof(TO) == 4) {
            *reinterpret_cast<float4*>(dx + e) = acc;
        } else {
            const bf16_t o[4] = {(bf16_t)acc.x, (bf16_t)acc.y, (bf16_t)acc.z, (bf16_t)acc.w};
            *reinterpret_cast<uint2*>(dx + e) = *reinterpret_cast<const uint2*>(o);
        }
    }
}

__global__ void dcn_flag_reset_kernel(int* __restrict__ flag) { *flag = 0; }

// flag (may be null): bit 1 set by an accumulating kernel that met a non-finite contribution -- the fixed-point conversion saturates those to finite
// garbage, so the whole input gradient is written as NaN instead (what a chain of fp32 atomics would have spread; the window route does the same)
template <class TO>
__global__ __launch_bounds__(256) void dcn_unfix_kernel(const unsigned long long* __restrict__ acc, TO* __restrict__ dx, int64_t n,
                                                        const int* __restrict__ flag = nullptr) {
    const bool poisoned = flag != nullptr && (*flag & 2) != 0;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = from_f32<TO>(poisoned ? __builtin_nanf("") : (float)((double)(long long)acc[i] * DCN_UNFIX));
}

__global__ __launch_bounds__(256) void dcn_zero_kernel(uint4* __restrict__ p, int64_t n16) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// dgamma partials of a per-channel scale y = x * gamma[c]:  out[c] = sum_r a[r][c] * b[r][c]
template <class T>
__global__ __launch_bounds__(256) void mul_colsum_partial_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t rows, int C,
                                                                 float* __restrict__ partials) {
    // block handles a contiguous row range; thread c-strides the channels (coalesced), sums its rows in order
    const int64_t rpb = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int64_t r = r0; r < r1; ++r) s = fmaf(to_f32(a[r * C + c]), to_f32(b[r * C + c]), s);
        partials[(int64_t)blockIdx.x * C + c] = s;
    }
}

// the same for C % 8 == 0: a lane owns an 8-channel chunk and every (256 / chunks)-th row, four row loads of both operands in
// flight per trip; lanes of a chunk meet through one LDS slab row per row lane summed in row order (no float atomics), blocks through
// the fixed-order partial reduction.  The scalar kernel above ran 56 us per call on InternImage-B (66 calls per step).
template <class T>
__global__ __launch_bounds__(256) void mul_colsum_partial_vec_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t rows,
                                                                     int C, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float lds_mc[];  // [rows per iteration][C]
    const int nch = C / 8;
    const int tpc = nch < 256 ? nch : 256;
    const int rpi = 256 / tpc;
    const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
    if (tr < rpi) {
        for (int c = tc; c < nch; c += tpc) {
            float s[8] = {};
            int64_t r = (int64_t)blockIdx.x * rpi + tr;
            const int64_t rstep = (int64_t)gridDim.x * rpi;
            constexpr int UB = 4;
            for (; r + (UB - 1) * rstep < rows; r += UB * rstep) {
                float va[UB][8], vb[UB][8];
#pragma unroll
                for (int q = 0; q < UB; ++q) {
                    load8<T>(a + (r + q * rstep) * C + c * 8, va[q]);
                    load8<T>(b + (r + q * rstep) * C + c * 8, vb[q]);
                }
#pragma unroll
                for (int q = 0; q < UB; ++q)
#pragma unroll
                    for (int u = 0; u < 8; ++u) s[u] = fmaf(va[q][u], vb[q][u], s[u]);
            }
            for (; r < rows; r += rstep) {
                float va[8], vb[8];
                load8<T>(a + r * C + c * 8, va);
                load8<T>(b + r * C + c * 8, vb);
#pragma unroll
                for (int u = 0; u < 8; ++u) s[u] = fmaf(va[u], vb[u], s[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) lds_mc[(size_t)tr * C + c * 8 + u] = s[u];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
        float a = lds_mc[i];
        for (int t = 1; t < rpi; ++t) a += lds_mc[(size_t)t * C + i];
        partials[(int64_t)blockIdx.x * C + i] = a;
    }
}

static size_t mc_slab_bytes(int C) {
    const int nch = C / 8;
    const int tpc = nch < 256 ? nch : 256;
    return (size_t)(256 / tpc) * C * sizeof(float);
}

// y[r][c] = x[r][c] * s[c]   (per-channel layer scale in the storage dtype)
template <class T>
__global__ __launch_bounds__(256) void scale_cols_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y,
                                                         int64_t n, int C) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = from_f32<T>(to_f32(x[i]) * s[i % C]);
}

template <class T>
__global__ __launch_bounds__(256) void scale_cols_vec_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y,
                                                             int64_t nchunks, int C) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * 256) {
        float v[8], g[8];
        load8<T>(x + i * 8, v);
        load8<float>(s + (i * 8) % C, g);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] *= g[u];
        store8<T>(y + i * 8, v);
    }
}

static inline int mc_blocks(int64_t rows) {
    int64_t b = ceil_div64(rows, 64);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

static inline unsigned lane_blocks(int64_t n) {
    int64_t b = ceil_div64(n, 256);
    if (b > 256 * 64) b = 256 * 64;
    if (b < 1) b = 1;
    return (unsigned)b;
}

static int make_geom(DcnGeom* g, int N, int H, int W, int G, int Cg, int kh, int kw, int stride, int dil, int pad, float s,
                     const char* who) {
    ISEG_REQUIRE(N > 0 && H > 0 && W > 0 && G > 0 && Cg > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0 && pad >= 0,
                 "%s: bad geometry", who);
    g->N = N; g->H = H; g->W = W; g->G = G; g->Cg = Cg; g->kh = kh; g->kw = kw; g->stride = stride; g->dil = dil; g->pad = pad;
    g->Hin = H + 2 * pad;
    g->Win = W + 2 * pad;
    g->Ho = (g->Hin - (dil * (kh - 1) + 1)) / stride + 1;
    g->Wo = (g->Win - (dil * (kw - 1) + 1)) / stride + 1;
    g->s = s;
    g->ld_off = 2 * G * kh * kw;
    g->ld_mask = G * kh * kw;
    ISEG_REQUIRE(g->Ho > 0 && g->Wo > 0, "%s: empty output", who);
    ISEG_REQUIRE((int64_t)N * g->Hin * g->Win * G * Cg < (1ll << 40), "%s: tensor too large", who);
    return ISEG_OK;
}


// window geometry of the deterministic backward pass; ok = 0 when the zero-offset footprint of a tile does not fit the window
static bool dcn_window(const DcnGeom& g, DcnWin* wn) {
    if (g.Cg != 8 && g.Cg != 16) return false;
    if (g.kh * g.kw > DCN_PMAX) return false;
    if (g.Win <= 2 || g.Hin <= 2) return false;
    // 32-bit index arithmetic of the gather kernel: element count and bracket products
    if ((int64_t)g.N * g.H * g.W * g.G * g.Cg >= (1ll << 31) || (int64_t)(g.Win + DCN_WS + 64) * g.Hin >= (1ll << 30) ||
        (int64_t)(g.Hin + DCN_WS + 64) * g.Win >= (1ll << 30) || (int64_t)DCN_TS * g.stride * (g.Win + g.Hin) >= (1ll << 30))
        return false;
    wn->tiles_y = (g.Ho + DCN_TS - 1) / DCN_TS;
    wn->tiles_x = (g.Wo + DCN_TS - 1) / DCN_TS;
    const double half = (double)((g.dil * (g.kh - 1)) / 2), halfw = (double)((g.dil * (g.kw - 1)) / 2);
    // px(h, j) = ((half + 0.5 + h stride) / Hin + (-halfw + j dil) s / Win) (Win - 2),  j = p / kh in [0, P / kh)
    // py(w, i) = ((halfw + 0.5 + w stride) / Win + (-half + i dil) s / Hin) (Hin - 2),  i = p % kh
    const int nj = (g.kh * g.kw + g.kh - 1) / g.kh, ni = g.kh;
    const double ax = (double)g.stride * (g.Win - 2) / g.Hin, ay = (double)g.stride * (g.Hin - 2) / g.Win;
    const double gx_lo = -halfw * g.s * (g.Win - 2) / g.Win, gx_hi = (-halfw + (nj - 1) * g.dil) * g.s * (g.Win - 2) / g.Win;
    const double gy_lo = -half * g.s * (g.Hin - 2) / g.Hin, gy_hi = (-half + (ni - 1) * g.dil) * g.s * (g.Hin - 2) / g.Hin;
    const double bx = (half + 0.5) * (g.Win - 2) / g.Hin, by = (halfw + 0.5) * (g.Hin - 2) / g.Win;
    const double lox = bx + (gx_lo < gx_hi ? gx_lo : gx_hi), hix = bx + (gx_lo < gx_hi ? gx_hi : gx_lo);
    const double loy = by + (gy_lo < gy_hi ? gy_lo : gy_hi), hiy = by + (gy_lo < gy_hi ? gy_hi : gy_lo);
    // corners x0 = floor(px), x1 = x0 + 1; the integer tile origin (ty TS stride (Win-2)) / Hin is within one pixel of ty TS ax
    const int spanx = (int)ceil((DCN_TS - 1) * ax + (hix - lox)) + 3, spany = (int)ceil((DCN_TS - 1) * ay + (hiy - loy)) + 3;
    if (spanx > DCN_WS || spany > DCN_WS) return false;
    wn->rx = (DCN_WS - spanx) / 2;
    wn->ry = (DCN_WS - spany) / 2;
    wn->c0x = (int)floor(lox) - 1;
    wn->c0y = (int)floor(loy) - 1;
    return true;
}

static size_t dcn_win_bytes(const DcnGeom& g, const DcnWin& wn, size_t* side_off, size_t* flag_off) {
    size_t win = (size_t)g.N * wn.tiles_y * wn.tiles_x * g.G * DCN_WS * DCN_WS * g.Cg * sizeof(float);
    win = (win + 255) / 256 * 256;
    size_t side = (size_t)g.N * g.H * g.W * g.G * g.Cg * sizeof(unsigned long long);
    side = (side + 255) / 256 * 256;
    *side_off = win;
    *flag_off = win + side;
    return win + side + 256;
}
}  // namespace

// row pitches of the offset / mask operands (and of their gradients): 0 = dense; otherwise >= the dense run, ld_off even (bf16 gradient pairs are
// stored as one dword)
static int set_pitches(DcnGeom* g, int64_t ld_off, int64_t ld_mask, const char* who) {
    if (ld_off == 0) ld_off = g->ld_off;
    if (ld_mask == 0) ld_mask = g->ld_mask;
    ISEG_REQUIRE(ld_off >= g->ld_off && ld_mask >= g->ld_mask && ld_off % 2 == 0 && ld_off < (1 << 30) && ld_mask < (1 << 30),
                 "%s: offset / mask row pitches %lld / %lld (dense runs %d / %d; the offset pitch must be even)", who, (long long)ld_off,
                 (long long)ld_mask, g->ld_off, g->ld_mask);
    g->ld_off = (int)ld_off;
    g->ld_mask = (int)ld_mask;
    return ISEG_OK;
}

extern "C" int iseg_dcnv3_fwd(const void* x, const void* offset, const void* mask, void* y, int N, int H, int W, int G, int Cg, int kh,
                              int kw, int stride, int dil, int pad, float offset_scale, int dtype, hipStream_t stream) {
    return iseg_dcnv3_fwd_ld(x, offset, mask, 0, 0, y, N, H, W, G, Cg, kh, kw, stride, dil, pad, offset_scale, dtype, stream);
}

extern "C" int iseg_dcnv3_fwd_ld(const void* x, const void* offset, const void* mask, int64_t ld_off, int64_t ld_mask, void* y, int N, int H, int W,
                                 int G, int Cg, int kh, int kw, int stride, int dil, int pad, float offset_scale, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && offset && mask && y, "iseg_dcnv3_fwd: null pointer");
    DcnGeom g;
    int rc = make_geom(&g, N, H, W, G, Cg, kh, kw, stride, dil, pad, offset_scale, "iseg_dcnv3_fwd");
    if (rc != ISEG_OK) return rc;
    rc = set_pitches(&g, ld_off, ld_mask, "iseg_dcnv3_fwd_ld");
    if (rc != ISEG_OK) return rc;
    const int64_t lanes = (int64_t)N * g.Ho * g.Wo * G;
    const bool v8 = Cg % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0;
#define DCN_FWD(T, CV)                                                                                                              \
    hipLaunchKernelGGL((dcnv3_fwd_kernel<T, CV>), dim3(lane_blocks(lanes)), dim3(256), 0, stream, (const T*)x, (const T*)offset,   \
                       (const T*)mask, (T*)y, g)
#define DCN_FWD_PIPE(T, CG)                                                                                                         \
    hipLaunchKernelGGL((dcnv3_fwd_pipe_kernel<T, CG>), dim3(lane_blocks(lanes)), dim3(256), 0, stream, (const T*)x, (const T*)offset, \
                       (const T*)mask, (T*)y, g)
    static const bool allow_pipe = [] { const char* e = getenv("ISEG_DCN_FWD_PIPE"); return !e || atoi(e) != 0; }();
    const bool pipe = allow_pipe && v8 && (Cg == 8 || Cg == 16) && kh * kw <= DCN_PMAX;
    if (dtype == ISEG_BF16) {
        if (pipe && Cg == 16) DCN_FWD_PIPE(bf16_t, 16);
        else if (pipe) DCN_FWD_PIPE(bf16_t, 8);
        else if (v8) DCN_FWD(bf16_t, 8);
        else DCN_FWD(bf16_t, 1);
    } else {
        if (pipe && Cg == 16) DCN_FWD_PIPE(float, 16);
        else if (pipe) DCN_FWD_PIPE(float, 8);
        else if (v8) DCN_FWD(float, 8);
        else DCN_FWD(float, 1);
    }
#undef DCN_FWD_PIPE
#undef DCN_FWD
    return iseg_check_launch("iseg_dcnv3_fwd");
}

// general route: nel int64 accumulators (rounded up to 16 bytes) + a 16-byte slot whose first word is the non-finite flag
static size_t dcn_general_bytes(int64_t nel) { return ((size_t)nel * sizeof(unsigned long long) + 15) / 16 * 16 + 16; }

extern "C" size_t iseg_dcnv3_bwd_workspace_bytes(int N, int H, int W, int G, int Cg, int kh, int kw, int stride, int dil, int pad,
                                                 float offset_scale) {
    DcnGeom g;
    if (make_geom(&g, N, H, W, G, Cg, kh, kw, stride, dil, pad, offset_scale, "iseg_dcnv3_bwd_workspace_bytes") != ISEG_OK) return 0;
    DcnWin wn;
    const size_t fallback = dcn_general_bytes((int64_t)N * H * W * G * Cg);      // int64 accumulators of the general kernels + their flag word
    if (!dcn_window(g, &wn)) return fallback;
    size_t so, fo;
    const size_t win = dcn_win_bytes(g, wn, &so, &fo);
    return win > fallback ? win : fallback;      // (ISEG_DCN_BWD_WIN=0 sends window geometries down the general route too)
}

extern "C" int iseg_dcnv3_bwd(const void* x, const void* offset, const void* mask, const void* dy, float* dx_f32, void* doffset,
                              void* dmask, int N, int H, int W, int G, int Cg, int kh, int kw, int stride, int dil, int pad,
                              float offset_scale, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
    return iseg_dcnv3_bwd_ld(x, offset, mask, 0, 0, dy, dx_f32, ISEG_F32, doffset, dmask, N, H, W, G, Cg, kh, kw, stride, dil, pad, offset_scale, dtype,
                             ws, ws_bytes, nullptr, 0, stream);
}

// bytes of the caller-kept side buffer of iseg_dcnv3_bwd_ld (0: this geometry does not take the window route)
extern "C" size_t iseg_dcnv3_bwd_side_bytes(int N, int H, int W, int G, int Cg, int kh, int kw, int stride, int dil, int pad, float offset_scale) {
    DcnGeom g;
    if (make_geom(&g, N, H, W, G, Cg, kh, kw, stride, dil, pad, offset_scale, "iseg_dcnv3_bwd_side_bytes") != ISEG_OK) return 0;
    DcnWin wn;
    if (!dcn_window(g, &wn)) return 0;
    size_t so, fo;
    const size_t all = dcn_win_bytes(g, wn, &so, &fo);
    return all - so;
}

extern "C" int iseg_dcnv3_bwd_ld(const void* x, const void* offset, const void* mask, int64_t ld_off, int64_t ld_mask, const void* dy, void* dx,
                                 int dx_dtype, void* doffset, void* dmask, int N, int H, int W, int G, int Cg, int kh, int kw, int stride, int dil,
                                 int pad, float offset_scale, int dtype, void* ws, size_t ws_bytes, void* side_keep, size_t side_keep_bytes,
                                 hipStream_t stream) {
    void* const dx_f32 = dx;
    ISEG_REQUIRE(x && offset && mask && dy && dx_f32 && doffset && dmask, "iseg_dcnv3_bwd: null pointer");
    ISEG_REQUIRE(dx_dtype == ISEG_F32 || dx_dtype == ISEG_BF16, "iseg_dcnv3_bwd_ld: dx_dtype %d", dx_dtype);
    DcnGeom g;
    int rc = make_geom(&g, N, H, W, G, Cg, kh, kw, stride, dil, pad, offset_scale, "iseg_dcnv3_bwd");
    if (rc != ISEG_OK) return rc;
    rc = set_pitches(&g, ld_off, ld_mask, "iseg_dcnv3_bwd_ld");
    if (rc != ISEG_OK) return rc;
    ISEG_REQUIRE(dtype != ISEG_BF16 || (uintptr_t)doffset % 4 == 0, "iseg_dcnv3_bwd_ld: doffset must be 4-byte aligned");
    DcnWin wn;
    static const bool allow_win = [] { const char* e = getenv("ISEG_DCN_BWD_WIN"); return !e || atoi(e) != 0; }();
    if (allow_win && dcn_window(g, &wn)) {
        size_t side_off, flag_off;
        const size_t need = dcn_win_bytes(g, wn, &side_off, &flag_off);
        if (!ws || ws_bytes < need) {
            iseg_set_error("iseg_dcnv3_bwd: needs %zu workspace bytes, got %zu", need, ws_bytes);
            return ISEG_ERR_WORKSPACE;
        }
        float* windows = (float*)ws;
        unsigned long long* side = (unsigned long long*)((char*)ws + side_off);
        int* flag = (int*)((char*)ws + flag_off);
        const bool keep = side_keep != nullptr;
        if (keep) {      // all zero on entry (the caller's promise), all zero again when the gather kernel and the flag reset have run
            ISEG_REQUIRE(side_keep_bytes >= need - side_off && (uintptr_t)side_keep % 16 == 0, "iseg_dcnv3_bwd_ld: the kept side buffer needs %zu bytes, got %zu",
                         need - side_off, side_keep_bytes);
            side = (unsigned long long*)side_keep;
            flag = (int*)((char*)side_keep + (flag_off - side_off));
        } else {
            const int64_t n16 = (int64_t)(need - side_off) / 16;      // side buffer + flag
            hipLaunchKernelGGL(dcn_zero_kernel, dim3(lane_blocks(n16)), dim3(256), 0, stream, (uint4*)side, n16);
        }
        const unsigned blocks = (unsigned)((int64_t)N * wn.tiles_y * wn.tiles_x * G);
        const size_t cells = Cg == 16 ? DcnWinLds<16>::CELLS : DcnWinLds<8>::CELLS;
        const size_t stage_bytes = (size_t)256 * 3 * DCN_PMAX * (dtype == ISEG_BF16 ? 2 : 4);      // staged mask / offset gradients
        const size_t lds = cells * (dtype == ISEG_BF16 ? sizeof(int) : sizeof(unsigned long long)) + stage_bytes;
#define DCN_WIN(T, CG)                                                                                                                      \
    do {                                                                                                                                    \
        static const bool raised = [] {                                                                                                     \
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&dcnv3_bwd_win_kernel<T, CG>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       DcnWinLds<CG>::CELLS * 8 + 256 * 3 * DCN_PMAX * 4) == hipSuccess;                                    \
        }();                                                                                                                                \
        (void)raised;                                                                                                                       \
        hipLaunchKernelGGL((dcnv3_bwd_win_kernel<T, CG>), dim3(blocks), dim3(256), lds, stream, (const T*)x, (const T*)offset, (const T*)mask, \
                           (const T*)dy, (T*)doffset, (T*)dmask, windows, side, flag, g, wn);                                               \
    } while (0)
        if (dtype == ISEG_BF16) {
            if (Cg == 16) DCN_WIN(bf16_t, 16);
            else DCN_WIN(bf16_t, 8);
        } else {
            if (Cg == 16) DCN_WIN(float, 16);
            else DCN_WIN(float, 8);
        }
#undef DCN_WIN
        const int64_t threads = (int64_t)N * H * W * G * (Cg / 4);
#define DCN_GATHER(CG, TO, RESET)                                                                                                          \
    hipLaunchKernelGGL((dcnv3_bwd_gather_kernel<CG, TO, RESET>), dim3(lane_blocks(threads)), dim3(256), 0, stream, windows, side, flag, (TO*)dx, g, wn)
        if (dx_dtype == ISEG_BF16) {
            if (Cg == 16) { if (keep) DCN_GATHER(16, bf16_t, true); else DCN_GATHER(16, bf16_t, false); }
            else { if (keep) DCN_GATHER(8, bf16_t, true); else DCN_GATHER(8, bf16_t, false); }
        } else {
            if (Cg == 16) { if (keep) DCN_GATHER(16, float, true); else DCN_GATHER(16, float, false); }
            else { if (keep) DCN_GATHER(8, float, true); else DCN_GATHER(8, float, false); }
        }
#undef DCN_GATHER
        if (keep) hipLaunchKernelGGL(dcn_flag_reset_kernel, dim3(1), dim3(1), 0, stream, flag);
        return iseg_check_launch("iseg_dcnv3_bwd");
    }
    // general route (other group widths, footprints wider than the window): int64 fixed-point atomics into a zeroed workspace, then one
    // conversion pass -- order-free like the window kernels
    const int64_t nel = (int64_t)N * H * W * G * Cg;
    const size_t need = dcn_general_bytes(nel);
    {
        if (!ws || ws_bytes < need) {
            iseg_set_error("iseg_dcnv3_bwd: needs %zu workspace bytes, got %zu", need, ws_bytes);
            return ISEG_ERR_WORKSPACE;
        }
        const int64_t n16 = (int64_t)need / 16;      // (a multiple of 16: accumulators and the flag slot are zeroed by one launch)
        hipLaunchKernelGGL(dcn_zero_kernel, dim3(lane_blocks(n16)), dim3(256), 0, stream, (uint4*)ws, n16);
    }
    unsigned long long* const acc = (unsigned long long*)ws;
    int* const gflag = (int*)((char*)ws + need - 16);
    const int64_t lanes = (int64_t)N * g.Ho * g.Wo * G;
    const bool v8 = Cg % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0;
#define DCN_BWD(T, CV)                                                                                                              \
    hipLaunchKernelGGL((dcnv3_bwd_kernel<T, CV>), dim3(lane_blocks(lanes)), dim3(256), 0, stream, (const T*)x, (const T*)offset,   \
                       (const T*)mask, (const T*)dy, acc, (T*)doffset, (T*)dmask, gflag, g)
#define DCN_BWD_CL(T, LC)                                                                                                            \
    hipLaunchKernelGGL((dcnv3_bwd_cl_kernel<T, LC>), dim3(lane_blocks(lanes * LC)), dim3(256), 0, stream, (const T*)x,              \
                       (const T*)offset, (const T*)mask, (const T*)dy, acc, (T*)doffset, (T*)dmask, gflag, g)
#define DCN_BWD_CL_ANY(T)                    \
    do {                                     \
        if (Cg == 4) DCN_BWD_CL(T, 4);       \
        else if (Cg == 8) DCN_BWD_CL(T, 8);  \
        else if (Cg == 16) DCN_BWD_CL(T, 16); \
        else if (Cg == 32) DCN_BWD_CL(T, 32); \
        else DCN_BWD_CL(T, 64);              \
    } while (0)
    const bool pow2 = Cg == 4 || Cg == 8 || Cg == 16 || Cg == 32 || Cg == 64;
    if (dtype == ISEG_BF16) {
        if (pow2) DCN_BWD_CL_ANY(bf16_t);
        else if (v8) DCN_BWD(bf16_t, 8);
        else DCN_BWD(bf16_t, 1);
    } else {
        if (pow2) DCN_BWD_CL_ANY(float);
        else if (v8) DCN_BWD(float, 8);
        else DCN_BWD(float, 1);
    }
#undef DCN_BWD_CL_ANY
#undef DCN_BWD_CL
#undef DCN_BWD
    if (dx_dtype == ISEG_BF16)
        hipLaunchKernelGGL(dcn_unfix_kernel<bf16_t>, dim3(lane_blocks(nel)), dim3(256), 0, stream, acc, (bf16_t*)dx, nel, (const int*)gflag);
    else hipLaunchKernelGGL(dcn_unfix_kernel<float>, dim3(lane_blocks(nel)), dim3(256), 0, stream, acc, (float*)dx, nel, (const int*)gflag);
    return iseg_check_launch("iseg_dcnv3_bwd");
}

// centre-feature scale of the DCNv3 layer (layers/dcn_v3/dcn_v3.py:138-146): out = x (1 - s) + x_proj s with one scale per (pixel, group),
// broadcast over the group's channels.  A lane owns 8 channels of one (pixel, group); the backward's ds = sum_c dout (x_proj - x) meets the
// other lanes of a 16-channel group through one DPP-free shuffle (Cg in {8, 16}).
template <class T, int CG>
__global__ __launch_bounds__(256) void dcn_center_blend_fwd_kernel(const T* __restrict__ x, const T* __restrict__ xp, const T* __restrict__ sc,
                                                                   T* __restrict__ out, int64_t groups_total) {
    constexpr int L = CG / 8;
    const int64_t total = groups_total * L;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float s = to_f32(sc[i / L]);
        float a[8], b[8];
        load8<T>(x + i * 8, a);
        load8<T>(xp + i * 8, b);
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = fmaf(b[u] - a[u], s, a[u]);
        store8<T>(out + i * 8, a);
    }
}

template <class T, int CG>
__global__ __launch_bounds__(256) void dcn_center_blend_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ xp,
                                                                   const T* __restrict__ sc, T* __restrict__ dx, T* __restrict__ dxp,
                                                                   T* __restrict__ ds, int64_t groups_total) {
    constexpr int L = CG / 8;
    const int64_t total = groups_total * L;
    const int64_t padded = (total + 255) / 256 * 256;      // every lane of a wavefront takes part in the shuffle
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < padded; i += (int64_t)gridDim.x * 256) {
        const bool live = i < total;
        const int64_t j = live ? i : 0;
        const float s = to_f32(sc[j / L]);
        float d[8], a[8], b[8], ga[8], gb[8];
        load8<T>(dout + j * 8, d);
        load8<T>(x + j * 8, a);
        load8<T>(xp + j * 8, b);
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            dot = fmaf(d[u], b[u] - a[u], dot);
            ga[u] = d[u] * (1.f - s);
            gb[u] = d[u] * s;
        }
        if (L == 2) dot += __shfl_xor(dot, 1);
        if (live) {
            store8<T>(dx + j * 8, ga);
            store8<T>(dxp + j * 8, gb);
            if (j % L == 0) ds[j / L] = from_f32<T>(dot);
        }
    }
}

extern "C" int iseg_dcn_center_blend_fwd(const void* x, const void* x_proj, const void* scale, void* out, int64_t pixels, int G, int Cg, int dtype,
                                         hipStream_t stream) {
    ISEG_REQUIRE(x && x_proj && scale && out && pixels > 0 && G > 0, "iseg_dcn_center_blend_fwd: bad arguments");
    ISEG_REQUIRE(Cg == 8 || Cg == 16, "iseg_dcn_center_blend_fwd: group width %d (8 or 16)", Cg);
    const int64_t groups_total = pixels * G;
#define DCN_BLEND(T, CG)                                                                                                                    \
    hipLaunchKernelGGL((dcn_center_blend_fwd_kernel<T, CG>), dim3(lane_blocks(groups_total * (CG / 8))), dim3(256), 0, stream, (const T*)x, \
                       (const T*)x_proj, (const T*)scale, (T*)out, groups_total)
    if (dtype == ISEG_BF16) {
        if (Cg == 16) DCN_BLEND(bf16_t, 16);
        else DCN_BLEND(bf16_t, 8);
    } else {
        if (Cg == 16) DCN_BLEND(float, 16);
        else DCN_BLEND(float, 8);
    }
#undef DCN_BLEND
    return iseg_check_launch("iseg_dcn_center_blend_fwd");
}

extern "C" int iseg_dcn_center_blend_bwd(const void* dout, const void* x, const void* x_proj, const void* scale, void* dx, void* dx_proj,
                                         void* dscale, int64_t pixels, int G, int Cg, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(dout && x && x_proj && scale && dx && dx_proj && dscale && pixels > 0 && G > 0, "iseg_dcn_center_blend_bwd: bad arguments");
    ISEG_REQUIRE(Cg == 8 || Cg == 16, "iseg_dcn_center_blend_bwd: group width %d (8 or 16)", Cg);
    const int64_t groups_total = pixels * G;
#define DCN_BLEND(T, CG)                                                                                                                     \
    hipLaunchKernelGGL((dcn_center_blend_bwd_kernel<T, CG>), dim3(lane_blocks(groups_total * (CG / 8))), dim3(256), 0, stream, (const T*)dout, \
                       (const T*)x, (const T*)x_proj, (const T*)scale, (T*)dx, (T*)dx_proj, (T*)dscale, groups_total)
    if (dtype == ISEG_BF16) {
        if (Cg == 16) DCN_BLEND(bf16_t, 16);
        else DCN_BLEND(bf16_t, 8);
    } else {
        if (Cg == 16) DCN_BLEND(float, 16);
        else DCN_BLEND(float, 8);
    }
#undef DCN_BLEND
    return iseg_check_launch("iseg_dcn_center_blend_bwd");
}

extern "C" size_t iseg_mul_colsum_workspace_bytes(int64_t rows, int C) { return (size_t)mc_blocks(rows) * C * sizeof(float); }

extern "C" int iseg_mul_colsum(const void* a, const void* b, int64_t rows, int C, float* out, int accumulate, int dtype, void* ws,
                               size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(a && b && out && rows > 0 && C > 0, "iseg_mul_colsum: bad arguments");
    int blocks = mc_blocks(rows);
    if (C % 8 == 0 && ((uintptr_t)a | (uintptr_t)b) % 16 == 0 && blocks > 256) blocks = 256;      // vector kernel: one round of blocks
    const size_t need = (size_t)blocks * C * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_mul_colsum: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const bool vec = C % 8 == 0 && ((uintptr_t)a | (uintptr_t)b) % 16 == 0;
    if (vec && dtype == ISEG_BF16)
        hipLaunchKernelGGL((mul_colsum_partial_vec_kernel<bf16_t>), dim3(blocks), dim3(256), mc_slab_bytes(C), stream,
                           (const bf16_t*)a, (const bf16_t*)b, rows, C, (float*)ws);
    else if (vec)
        hipLaunchKernelGGL((mul_colsum_partial_vec_kernel<float>), dim3(blocks), dim3(256), mc_slab_bytes(C), stream,
                           (const float*)a, (const float*)b, rows, C, (float*)ws);
    else if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((mul_colsum_partial_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b,
                           rows, C, (float*)ws);
    else
        hipLaunchKernelGGL((mul_colsum_partial_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)a, (const float*)b, rows,
                           C, (float*)ws);
    launch_reduce_rows((const float*)ws, blocks, C, 0, 1, C, out, nullptr, C, 0, 1.f, accumulate, stream);
    return iseg_check_launch("iseg_mul_colsum");
}

extern "C" int iseg_scale_cols(const void* x, const float* colscale, void* y, int64_t rows, int C, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && colscale && y && rows > 0 && C > 0, "iseg_scale_cols: bad arguments");
    const int64_t n = rows * C;
    const bool vec = C % 8 == 0 && ((uintptr_t)x | (uintptr_t)y) % 16 == 0 && (uintptr_t)colscale % 16 == 0;
    if (vec && dtype == ISEG_BF16)
        hipLaunchKernelGGL((scale_cols_vec_kernel<bf16_t>), dim3(lane_blocks(n / 8)), dim3(256), 0, stream, (const bf16_t*)x, colscale,
                           (bf16_t*)y, n / 8, C);
    else if (vec)
        hipLaunchKernelGGL((scale_cols_vec_kernel<float>), dim3(lane_blocks(n / 8)), dim3(256), 0, stream, (const float*)x, colscale,
                           (float*)y, n / 8, C);
    else if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((scale_cols_kernel<bf16_t>), dim3(lane_blocks(n)), dim3(256), 0, stream, (const bf16_t*)x, colscale, (bf16_t*)y,
                           n, C);
    else
        hipLaunchKernelGGL((scale_cols_kernel<float>), dim3(lane_blocks(n)), dim3(256), 0, stream, (const float*)x, colscale, (float*)y, n,
                           C);
    return iseg_check_launch("iseg_scale_cols");
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// DCNv2 (layers/dcn_v2.py:16-281 of the reference, the deformable convolution of FaPN's FeatureAlignment, layers/fapn.py:44-80): the sampling half.
//   off [N, H, W, 27] = the offset convolution's output: 9 x (dy, dx) then 9 mask logits (:114-137, deformable_groups = 1)
//   col [N, H, W, 9, C]: col[.., p, :] = sigmoid(logit_p) * bilinear(x zero-padded by 1, (h + ky + dy_p, w + kx + dx_p))     p = 3 ky + kx
// with the reference's arithmetic: the sampling position, its floor and floor + 1 are all clipped to [0, H + 1] x [0, W + 1] (padded coordinates), and
// the four weights come from the CLIPPED values (:150-189); the convolution itself is then one GEMM col [N H W, 9 C] x kernel [9 C, filters]
// (:230-240).  Coordinates and weights in fp32 whatever the storage type (as DCNv3 here).  Backward: d offset / d logit per (pixel, point) by a lane-group
// sum over the channels; dx by 64-bit fixed-point atomics (order-free, bit-reproducible) into a zeroed accumulator, converted by dcn_unfix_kernel.
// ---------------------------------------------------------------------------------------------------------------------------------------------
namespace {

struct Dcn2Tap {
    int y0, y1, x0, x1;                      // clipped corners, UNPADDED coordinates (-1 and H / W are the zero border)
    float d0y, d0x, d1y, d1x;                // gy - y0, gx - x0, y1 - gy, x1 - gx   (clipped values)
    float m, in_y, in_x;                     // sigmoid(logit); 1 where the unclipped position lies inside the clip range (tf.clip_by_value's gradient)
};

template <class T>
__device__ __forceinline__ Dcn2Tap dcn2_tap(const T* __restrict__ off, int h, int w, int p, int H, int W) {
    const float oy = to_f32(off[2 * p]), ox = to_f32(off[2 * p + 1]);
    const float gy_u = (float)(h + p / 3) + oy, gx_u = (float)(w + p % 3) + ox;      // (h + ph) + (ky - ph) + dy, padded coordinates
    const float hy = (float)(H + 1), hx = (float)(W + 1);
    Dcn2Tap t;
    t.in_y = gy_u >= 0.f && gy_u <= hy ? 1.f : 0.f;
    t.in_x = gx_u >= 0.f && gx_u <= hx ? 1.f : 0.f;
    const float fy = floorf(gy_u), fx = floorf(gx_u);
    const float y1 = fminf(fmaxf(fy + 1.f, 0.f), hy), x1 = fminf(fmaxf(fx + 1.f, 0.f), hx);
    const float y0 = fminf(fmaxf(fy, 0.f), hy), x0 = fminf(fmaxf(fx, 0.f), hx);
    const float gy = fminf(fmaxf(gy_u, 0.f), hy), gx = fminf(fmaxf(gx_u, 0.f), hx);
    t.d0y = gy - y0;
    t.d0x = gx - x0;
    t.d1y = y1 - gy;
    t.d1x = x1 - gx;
    t.y0 = (int)y0 - 1;
    t.y1 = (int)y1 - 1;
    t.x0 = (int)x0 - 1;
    t.x1 = (int)x1 - 1;
    t.m = 1.f / (1.f + expf(-to_f32(off[18 + p])));
    return t;
}

// one thread = (pixel, point, 8 channels)
template <class T>
__global__ __launch_bounds__(256) void dcnv2_sample_fwd_kernel(const T* __restrict__ x, const T* __restrict__ off, T* __restrict__ col, int N, int H, int W,
                                                               int C) {
    const int cv = C / 8;
    const int64_t total = (int64_t)N * H * W * 9 * cv;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * 8;
        int64_t t = i / cv;
        const int p = (int)(t % 9);
        const int64_t pix = t / 9;
        const int w = (int)(pix % W), h = (int)((pix / W) % H);
        const int64_t n = pix / ((int64_t)W * H);
        const Dcn2Tap tp = dcn2_tap(off + pix * 27, h, w, p, H, W);
        // corners in the reference's order (:167-173): (y1, x1), (y1, x0), (y0, x1), (y0, x0) with weights d0y d0x, d0y d1x, d1y d0x, d1y d1x
        const int ys[4] = {tp.y1, tp.y1, tp.y0, tp.y0}, xs[4] = {tp.x1, tp.x0, tp.x1, tp.x0};
        const float ws4[4] = {tp.d0y * tp.d0x, tp.d0y * tp.d1x, tp.d1y * tp.d0x, tp.d1y * tp.d1x};
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((unsigned)ys[k] < (unsigned)H && (unsigned)xs[k] < (unsigned)W) {
                float v[8];
                load8<T>(x + ((n * H + ys[k]) * W + xs[k]) * C + c, v);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = fmaf(ws4[k], v[u], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] *= tp.m;
        store8<T>(col + (pix * 9 + p) * C + c, acc);
    }
}

// LC lanes per (pixel, point): each walks its 8-channel chunks; d offset / d logit meet by a lane-group sum
template <class T, int LC>
__global__ __launch_bounds__(256) void dcnv2_sample_bwd_kernel(const T* __restrict__ x, const T* __restrict__ off, const T* __restrict__ dcol,
                                                               unsigned long long* __restrict__ dx, T* __restrict__ doff, int N, int H, int W, int C) {
    const int64_t total = (int64_t)N * H * W * 9 * LC;
    const int cv = C / 8;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < (total + 255) / 256 * 256; i += (int64_t)gridDim.x * 256) {
        const bool live = i < total;
        const int64_t ii = live ? i : total - 1;
        const int lc = (int)(ii % LC);
        const int64_t t = ii / LC;
        const int p = (int)(t % 9);
        const int64_t pix = t / 9;
        const int w = (int)(pix % W), h = (int)((pix / W) % H);
        const int64_t n = pix / ((int64_t)W * H);
        const Dcn2Tap tp = dcn2_tap(off + pix * 27, h, w, p, H, W);
        const int ys[4] = {tp.y1, tp.y1, tp.y0, tp.y0}, xs[4] = {tp.x1, tp.x0, tp.x1, tp.x0};
        const float ws4[4] = {tp.d0y * tp.d0x, tp.d0y * tp.d1x, tp.d1y * tp.d0x, tp.d1y * tp.d1x};
        // d bilinear / d gy = d0x (v11 - v01) + d1x (v10 - v00);  d / d gx = d0y (v11 - v10) + d1y (v01 - v00)
        const float wy[4] = {tp.d0x, tp.d1x, -tp.d0x, -tp.d1x}, wx[4] = {tp.d0y, -tp.d0y, tp.d1y, -tp.d1y};
        float g_m = 0.f, g_y = 0.f, g_x = 0.f;
        for (int cc = lc; cc < cv && live; cc += LC) {
            const int c = cc * 8;
            float d[8];
            load8<T>(dcol + (pix * 9 + p) * C + c, d);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((unsigned)ys[k] < (unsigned)H && (unsigned)xs[k] < (unsigned)W) {
                    const int64_t src = ((n * H + ys[k]) * W + xs[k]) * C + c;
                    float v[8];
                    load8<T>(x + src, v);
                    float dot = 0.f;
                    const float coef = tp.m * ws4[k] * DCN_FIX;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        dot = fmaf(d[u], v[u], dot);
                        atomicAdd(dx + src + u, dcn_to_fixed(d[u] * coef));
                    }
                    g_m = fmaf(ws4[k], dot, g_m);
                    g_y = fmaf(wy[k], dot, g_y);
                    g_x = fmaf(wx[k], dot, g_x);
                }
            }
        }
        g_m = group_sum(g_m, LC);
        g_y = group_sum(g_y, LC);
        g_x = group_sum(g_x, LC);
        if (live && lc == 0) {
            T* o = doff + pix * 27;
            o[2 * p] = from_f32<T>(g_y * tp.m * tp.in_y);
            o[2 * p + 1] = from_f32<T>(g_x * tp.m * tp.in_x);
            o[18 + p] = from_f32<T>(g_m * tp.m * (1.f - tp.m));
        }
    }
}

}  // namespace

extern "C" int iseg_dcnv2_sample_fwd(const void* x, const void* offset, void* col, int N, int H, int W, int C, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(x && offset && col && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "iseg_dcnv2_sample_fwd: bad arguments (C = %d must be a multiple of 8)", C);
    ISEG_REQUIRE((int64_t)N * H * W * 9 * C < (1ll << 40), "iseg_dcnv2_sample_fwd: tensor too large");
    const int64_t lanes = (int64_t)N * H * W * 9 * (C / 8);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((dcnv2_sample_fwd_kernel<bf16_t>), dim3(lane_blocks(lanes)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)offset, (bf16_t*)col,
                           N, H, W, C);
    else
        hipLaunchKernelGGL((dcnv2_sample_fwd_kernel<float>), dim3(lane_blocks(lanes)), dim3(256), 0, stream, (const float*)x, (const float*)offset, (float*)col, N,
                           H, W, C);
    return iseg_check_launch("iseg_dcnv2_sample_fwd");
}

extern "C" size_t iseg_dcnv2_sample_bwd_workspace_bytes(int N, int H, int W, int C) {
    return ((size_t)N * H * W * C * sizeof(unsigned long long) + 15) / 16 * 16;
}

extern "C" int iseg_dcnv2_sample_bwd(const void* x, const void* offset, const void* dcol, float* dx_f32, void* doffset, int N, int H, int W, int C,
                                     int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(x && offset && dcol && dx_f32 && doffset && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "iseg_dcnv2_sample_bwd: bad arguments");
    const size_t need = iseg_dcnv2_sample_bwd_workspace_bytes(N, H, W, C);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_dcnv2_sample_bwd: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const int64_t nel = (int64_t)N * H * W * C, n16 = (int64_t)(need / 16);
    unsigned long long* acc = (unsigned long long*)ws;
    hipLaunchKernelGGL(dcn_zero_kernel, dim3(lane_blocks(n16)), dim3(256), 0, stream, (uint4*)ws, n16);
    constexpr int LC = 8;
    const int64_t lanes = (int64_t)N * H * W * 9 * LC;
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((dcnv2_sample_bwd_kernel<bf16_t, LC>), dim3(lane_blocks(lanes)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)offset,
                           (const bf16_t*)dcol, acc, (bf16_t*)doffset, N, H, W, C);
    else
        hipLaunchKernelGGL((dcnv2_sample_bwd_kernel<float, LC>), dim3(lane_blocks(lanes)), dim3(256), 0, stream, (const float*)x, (const float*)offset,
                           (const float*)dcol, acc, (float*)doffset, N, H, W, C);
    hipLaunchKernelGGL(dcn_unfix_kernel<float>, dim3(lane_blocks(nel)), dim3(256), 0, stream, acc, dx_f32, nel, (const int*)nullptr);
    return iseg_check_launch("iseg_dcnv2_sample_bwd");
}


// ---------------------------------------------------------------------------------------------------------------------------
// The joint offset | mask projection of a DCNv3 layer (layers/dcn_v3/dcn_v3.py:116-123: two Dense layers on the same input, a softmax over each
// group's P mask logits) as ONE product into a [pixels][ld] buffer: columns [0, 2GP) offsets, [2GP, 3GP) mask, then padding up to ld (% 8 == 0, the
// LDS-DMA GEMM's output width).  The pieces around the GEMM:
//   softmax over the P mask entries of every (pixel, group), in place, rows `ld` apart;
//   its backward, which also clears the padding columns of the gradient buffer (they meet zero weights in the data-gradient product, and 0 x NaN
//   is NaN);
//   the weight / bias gradient of the joint product ([rows][ld] fp32) added column-range-wise into the two layers' gradients.
// ---------------------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void dcn_mask_softmax_fwd_kernel(T* __restrict__ om, int64_t pixels, int G, int P, int ld, int col0) {
    const int64_t total = pixels * G;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        T* r = om + (i / G) * ld + col0 + (int)(i % G) * P;
        float v[DCN_PMAX], mx = -3.0e38f;
#pragma unroll
        for (int p = 0; p < DCN_PMAX; ++p) {
            v[p] = p < P ? to_f32(r[p]) : -3.0e38f;
            mx = fmaxf(mx, v[p]);
        }
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < DCN_PMAX; ++p) {
            v[p] = p < P ? expf(v[p] - mx) : 0.f;
            sum += v[p];
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int p = 0; p < DCN_PMAX; ++p)
            if (p < P) r[p] = from_f32<T>(v[p] * inv);
    }
}

template <class T>
__global__ __launch_bounds__(256) void dcn_mask_softmax_bwd_kernel(const T* __restrict__ om, T* __restrict__ dom, int64_t pixels, int G, int P, int ld,
                                                                   int col0) {
    const int64_t total = pixels * G;
    const int pad0 = col0 + G * P;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / G;
        const int gi = (int)(i % G);
        const T* y = om + pix * ld + col0 + gi * P;
        T* d = dom + pix * ld + col0 + gi * P;
        float yv[DCN_PMAX], dv[DCN_PMAX], dot = 0.f;
#pragma unroll
        for (int p = 0; p < DCN_PMAX; ++p) {
            yv[p] = p < P ? to_f32(y[p]) : 0.f;
            dv[p] = p < P ? to_f32(d[p]) : 0.f;
            dot = fmaf(yv[p], dv[p], dot);
        }
#pragma unroll
        for (int p = 0; p < DCN_PMAX; ++p)
            if (p < P) d[p] = from_f32<T>(yv[p] * (dv[p] - dot));
        if (gi == 0)
            for (int c = pad0; c < ld; ++c) dom[pix * ld + c] = from_f32<T>(0.f);
    }
}

__global__ __launch_bounds__(256) void split_cols_accumulate_kernel(const float* __restrict__ src, int64_t rows, int ld, float* __restrict__ dst0, int n0,
                                                                    float* __restrict__ dst1, int n1) {
    const int64_t total = rows * (n0 + n1);
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (n0 + n1);
        const int c = (int)(i % (n0 + n1));
        const float v = src[r * ld + c];
        if (c < n0) {
            if (dst0) dst0[r * n0 + c] += v;
        } else if (dst1) {
            dst1[r * n1 + (c - n0)] += v;
        }
    }
}

extern "C" int iseg_dcn_mask_softmax_fwd(void* om, int64_t pixels, int G, int P, int64_t ld, int col0, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(om && pixels > 0 && G > 0 && P > 0 && P <= DCN_PMAX && col0 >= 0 && ld >= col0 + (int64_t)G * P && ld < (1 << 30),
                 "iseg_dcn_mask_softmax_fwd: bad arguments (P <= %d)", DCN_PMAX);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL(dcn_mask_softmax_fwd_kernel<bf16_t>, dim3(lane_blocks(pixels * G)), dim3(256), 0, stream, (bf16_t*)om, pixels, G, P, (int)ld, col0);
    else
        hipLaunchKernelGGL(dcn_mask_softmax_fwd_kernel<float>, dim3(lane_blocks(pixels * G)), dim3(256), 0, stream, (float*)om, pixels, G, P, (int)ld, col0);
    return iseg_check_launch("iseg_dcn_mask_softmax_fwd");
}

extern "C" int iseg_dcn_mask_softmax_bwd(const void* om, void* dom, int64_t pixels, int G, int P, int64_t ld, int col0, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(om && dom && pixels > 0 && G > 0 && P > 0 && P <= DCN_PMAX && col0 >= 0 && ld >= col0 + (int64_t)G * P && ld < (1 << 30),
                 "iseg_dcn_mask_softmax_bwd: bad arguments (P <= %d)", DCN_PMAX);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL(dcn_mask_softmax_bwd_kernel<bf16_t>, dim3(lane_blocks(pixels * G)), dim3(256), 0, stream, (const bf16_t*)om, (bf16_t*)dom, pixels, G,
                           P, (int)ld, col0);
    else
        hipLaunchKernelGGL(dcn_mask_softmax_bwd_kernel<float>, dim3(lane_blocks(pixels * G)), dim3(256), 0, stream, (const float*)om, (float*)dom, pixels, G, P,
                           (int)ld, col0);
    return iseg_check_launch("iseg_dcn_mask_softmax_bwd");
}

// dst0 [rows][n0] += src[:, 0:n0],  dst1 [rows][n1] += src[:, n0:n0+n1]   (src [rows][ld] fp32; a null destination is skipped)
extern "C" int iseg_split_cols_accumulate(const float* src, int64_t rows, int64_t ld, float* dst0, int n0, float* dst1, int n1, hipStream_t stream) {
    ISEG_REQUIRE(src && rows > 0 && n0 >= 0 && n1 >= 0 && n0 + n1 > 0 && ld >= n0 + n1 && ld < (1 << 30), "iseg_split_cols_accumulate: bad arguments");
    hipLaunchKernelGGL(split_cols_accumulate_kernel, dim3(lane_blocks(rows * (n0 + n1))), dim3(256), 0, stream, src, rows, (int)ld, dst0, n0, dst1, n1);
    return iseg_check_launch("iseg_split_cols_accumulate");
}
