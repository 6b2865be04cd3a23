// Global Response Normalization of ConvNeXt V2 (reference backbones/convnext_v2.py:17-60 GlobalResponseNormlizationLayer.call):
//   gx[n,c] = sqrt(sum_hw x^2 + eps),  nx[n,c] = gx / (mean_c gx + eps),  y = gamma * (x * nx) + beta + x      (fp32 arithmetic)
// on the [N, HW, C] hidden tensor of a block (C = 4 x filters).  HBM-bound: the forward reads x twice and writes y once, the backward reads
// dy and x twice and writes dx once; everything per (sample, channel) lives in [N, C] fp32 side arrays that stay in L2.
//   forward : grn_colsum<false> (sum x^2, per-workgroup partial rows) -> reduce_rows -> grn_stats -> grn_apply
//   backward: grn_colsum<true>  (sum dy*x | sum dy in ONE pass over dy) -> reduce_rows -> grn_bwd_stats -> reduce over samples (dgamma | dbeta,
//             deferred when the trainer's queue is open) -> grn_apply<bwd>
// Every reduction runs in a fixed order (no atomics): results are bit-reproducible.
#include "common.h"

namespace {

// partial column sums over a strided subset of the HW rows of sample blockIdx.y.  A lane owns one 8-channel chunk; `rpi` row lanes share a
// chunk column when C/8 < 256 and are combined through LDS in lane order.  DOT: out[0:C] = sum dy*x, out[C:2C] = sum dy; else out[0:C] = sum x*x.
template <class T, bool DOT>
__global__ __launch_bounds__(256) void grn_colsum_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partials,
                                                         int64_t HW, int C) {
    __shared__ __attribute__((aligned(16))) float red[DOT ? 2 : 1][256 * 8];
    const int nch = C / 8;
    const int tpc = nch < 256 ? nch : 256;
    const int rpi = 256 / tpc;
    const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
    const int64_t base = (int64_t)blockIdx.y * HW * C;
    constexpr int NO = DOT ? 2 : 1;
    float* out = partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (int64_t)(NO * C);
    const int64_t rstep = (int64_t)gridDim.x * rpi;
    for (int c0 = 0; c0 < nch; c0 += tpc) {
        const int c = c0 + tc;
        const bool live = tr < rpi && c < nch;
        float s[8] = {}, sd[8] = {};
        if (live) {
            int64_t r = (int64_t)blockIdx.x * rpi + tr;
            constexpr int UB = 4;      // four rows per trip, loads issued together
            for (; r + (UB - 1) * rstep < HW; r += UB * rstep) {
                float v[UB][8], d[DOT ? UB : 1][8];
#pragma unroll
                for (int q = 0; q < UB; ++q) {
                    load8<T>(x + base + (r + q * rstep) * C + c * 8, v[q]);
                    if (DOT) load8<T>(dy + base + (r + q * rstep) * C + c * 8, d[q]);
                }
#pragma unroll
                for (int q = 0; q < UB; ++q)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (DOT) {
                            s[u] = fmaf(d[q][u], v[q][u], s[u]);
                            sd[u] += d[q][u];
                        } else {
                            s[u] = fmaf(v[q][u], v[q][u], s[u]);
                        }
                    }
            }
            for (; r < HW; r += rstep) {
                float v[8], d[8];
                load8<T>(x + base + r * C + c * 8, v);
                if (DOT) load8<T>(dy + base + r * C + c * 8, d);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (DOT) {
                        s[u] = fmaf(d[u], v[u], s[u]);
                        sd[u] += d[u];
                    } else {
                        s[u] = fmaf(v[u], v[u], s[u]);
                    }
                }
            }
        }
        if (rpi > 1) {      // (uniform)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                red[0][threadIdx.x * 8 + u] = s[u];
                if (DOT) red[NO - 1][threadIdx.x * 8 + u] = sd[u];
            }
            __syncthreads();
            if (live && tr == 0) {
                for (int q = 1; q < rpi; ++q)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        s[u] += red[0][(q * tpc + tc) * 8 + u];
                        if (DOT) sd[u] += red[NO - 1][(q * tpc + tc) * 8 + u];
                    }
            }
            __syncthreads();
        }
        if (live && tr == 0) {
            store8<float>(out + c * 8, s);
            if (DOT) store8<float>(out + C + c * 8, sd);
        }
    }
}

// fixed-order sum over the 256 lanes of a workgroup (every lane gets the result)
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = group_sum(v, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per sample: gx = sqrt(sumsq + eps), nx = gx / (mean_c gx + eps)
__global__ __launch_bounds__(256) void grn_stats_kernel(const float* __restrict__ sumsq, float* __restrict__ nx, float* __restrict__ gx,
                                                        int C, float eps) {
    __shared__ float red[4];
    const int64_t o = (int64_t)blockIdx.x * C;
    float acc = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = sqrtf(sumsq[o + c] + eps);
        gx[o + c] = g;
        acc += g;
    }
    const float m = block_sum_256(acc, red) / (float)C + eps;
    for (int c = threadIdx.x; c < C; c += 256) nx[o + c] = gx[o + c] / m;      // (a lane re-reads what it wrote itself)
}

// one workgroup per sample.  buf[n][0:C] holds D = sum_hw dy*x on entry and D*nx (this sample's dgamma term) on exit;
// t[n,c] = d(loss)/d(gx) / gx, so that dx = dy*(gamma*nx + 1) + x*t.
__global__ __launch_bounds__(256) void grn_bwd_stats_kernel(float* __restrict__ buf, int64_t ldb, const float* __restrict__ gamma,
                                                            const float* __restrict__ nx, const float* __restrict__ gx,
                                                            float* __restrict__ t, int C, float eps) {
    __shared__ float red[4];
    const int64_t o = (int64_t)blockIdx.x * C;
    float* D = buf + (int64_t)blockIdx.x * ldb;
    float sg = 0.f, sr = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = gx[o + c];
        sg += g;
        sr = fmaf(gamma[c] * D[c], g, sr);
    }
    const float m = block_sum_256(sg, red) / (float)C + eps;
    const float r = block_sum_256(sr, red) / ((float)C * m * m);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float d = D[c];
        t[o + c] = (gamma[c] * d / m - r) / gx[o + c];
        D[c] = d * nx[o + c];
    }
}

// forward: y = x * (gamma*nx + 1) + beta;   backward (BWD): dx = (dy * (gamma*nx + 1) + x * t) [* mul]
// `mul` (optional, same shape as x) is the saved derivative of the activation that produced x: the chain rule through the GELU in front of the
// normalisation costs one more read here instead of a pass of its own
template <class T, bool BWD>
__global__ __launch_bounds__(256) void grn_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ nx,
                                                        const float* __restrict__ t, const T* __restrict__ mul, T* __restrict__ y, int64_t HW,
                                                        int C, int64_t chunks) {
    const int nch = C / 8;
    const int64_t per = HW * nch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / per);
        const int c = (int)(i % nch) * 8;
        float v[8], g[8], a[8], b[8], o[8];
        load8<T>(x + i * 8, v);
        load8<float>(gamma + c, g);
        load8<float>(nx + (int64_t)n * C + c, a);
        if (BWD) {
            float d[8];
            load8<T>(dy + i * 8, d);
            load8<float>(t + (int64_t)n * C + c, b);
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = fmaf(d[u], fmaf(g[u], a[u], 1.f), v[u] * b[u]);
            if (mul) {      // (uniform)
                float m[8];
                load8<T>(mul + i * 8, m);
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] *= m[u];
            }
        } else {
            load8<float>(beta + c, b);
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = (g[u] * (v[u] * a[u]) + b[u]) + v[u];      // gamma*(x*nx) + beta + x, the reference's order
        }
        store8<T>(y + i * 8, o);
    }
}

int grn_parts(int64_t N, int64_t HW, int C) {
    const int nch = C / 8;
    const int tpc = nch < 256 ? nch : 256;
    const int rpi = 256 / tpc;
    int64_t P = ceil_div64(HW, (int64_t)rpi * 8);      // >= 8 rows per lane
    const int64_t cap = N >= 2048 ? 1 : 2048 / N;      // ~2048 workgroups over the samples
    if (P > cap) P = cap;
    if (P > 256) P = 256;
    if (P < 1) P = 1;
    return (int)P;
}

size_t align256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

extern "C" size_t iseg_grn_workspace_bytes(int64_t N, int64_t HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    const size_t parts = (size_t)N * grn_parts(N, HW, C) * 2 * C * sizeof(float);
    return align256(parts) + 2 * align256((size_t)N * 2 * C * sizeof(float));
}

extern "C" int iseg_grn_fwd(const void* x, const float* gamma, const float* beta, void* y, float* nx, float* gx, int64_t N, int64_t HW,
                            int C, float eps, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(x && gamma && beta && y && nx && gx && N > 0 && HW > 0 && C > 0, "iseg_grn_fwd: bad arguments");
    ISEG_REQUIRE((dtype == ISEG_BF16 || dtype == ISEG_F32) && C % 8 == 0, "iseg_grn_fwd: C %% 8 == 0 required (got dtype %d, C %d)", dtype, C);
    ISEG_REQUIRE(N <= 65535, "iseg_grn_fwd: at most 65535 samples");
    ISEG_REQUIRE(ws && ws_bytes >= iseg_grn_workspace_bytes(N, HW, C), "iseg_grn_fwd: workspace too small");
    const int P = grn_parts(N, HW, C);
    float* parts = (float*)ws;
    float* sumsq = (float*)((char*)ws + align256((size_t)N * P * 2 * C * sizeof(float)));
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((grn_colsum_kernel<bf16_t, false>), dim3(P, (unsigned)N), dim3(256), 0, stream, (const bf16_t*)x,
                           (const bf16_t*)nullptr, parts, HW, C);
    else
        hipLaunchKernelGGL((grn_colsum_kernel<float, false>), dim3(P, (unsigned)N), dim3(256), 0, stream, (const float*)x, (const float*)nullptr,
                           parts, HW, C);
    launch_reduce_rows(parts, P, C, (int64_t)P * C, (int)N, C, sumsq, nullptr, C, C, 1.f, 0, stream);
    hipLaunchKernelGGL(grn_stats_kernel, dim3((unsigned)N), dim3(256), 0, stream, (const float*)sumsq, nx, gx, C, eps);
    const int64_t chunks = N * HW * (C / 8);
    const unsigned blocks = (unsigned)(ceil_div64(chunks, 256) < 8192 ? ceil_div64(chunks, 256) : 8192);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((grn_apply_kernel<bf16_t, false>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)nullptr,
                           gamma, beta, (const float*)nx, (const float*)nullptr, (const bf16_t*)nullptr, (bf16_t*)y, HW, C, chunks);
    else
        hipLaunchKernelGGL((grn_apply_kernel<float, false>), dim3(blocks), dim3(256), 0, stream, (const float*)x, (const float*)nullptr, gamma,
                           beta, (const float*)nx, (const float*)nullptr, (const float*)nullptr, (float*)y, HW, C, chunks);
    return iseg_check_launch("iseg_grn_fwd");
}

extern "C" int iseg_grn_bwd(const void* dy, const void* x, const float* gamma, const float* nx, const float* gx, const void* mul, void* dx,
                            float* dgamma, float* dbeta, int accumulate, int64_t N, int64_t HW, int C, float eps, int dtype, void* ws, size_t ws_bytes,
                            hipStream_t stream) {
    ISEG_REQUIRE(dy && x && gamma && nx && gx && dx && dgamma && dbeta && N > 0 && HW > 0 && C > 0, "iseg_grn_bwd: bad arguments");
    ISEG_REQUIRE((dtype == ISEG_BF16 || dtype == ISEG_F32) && C % 8 == 0, "iseg_grn_bwd: C %% 8 == 0 required (got dtype %d, C %d)", dtype, C);
    ISEG_REQUIRE(N <= 65535, "iseg_grn_bwd: at most 65535 samples");
    ISEG_REQUIRE(ws && ws_bytes >= iseg_grn_workspace_bytes(N, HW, C), "iseg_grn_bwd: workspace too small");
    const int P = grn_parts(N, HW, C);
    const size_t side = align256((size_t)N * 2 * C * sizeof(float));
    float* parts = (float*)ws;
    float* t = (float*)((char*)ws + align256((size_t)N * P * 2 * C * sizeof(float)));
    float* buf = (float*)((char*)t + side);
    // per-sample rows (D*nx | sum dy) are the partials of the parameter gradients: they go to the trainer's arena when its queue is open
    float* arena = iseg_deferred_partials((size_t)N * 2 * C * sizeof(float), dgamma, dbeta, accumulate, stream);
    if (arena) buf = arena;
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((grn_colsum_kernel<bf16_t, true>), dim3(P, (unsigned)N), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy,
                           parts, HW, C);
    else
        hipLaunchKernelGGL((grn_colsum_kernel<float, true>), dim3(P, (unsigned)N), dim3(256), 0, stream, (const float*)x, (const float*)dy, parts,
                           HW, C);
    launch_reduce_rows(parts, P, 2 * (int64_t)C, (int64_t)P * 2 * C, (int)N, 2 * (int64_t)C, buf, nullptr, 2 * (int64_t)C, 2 * (int64_t)C, 1.f,
                       0, stream);
    hipLaunchKernelGGL(grn_bwd_stats_kernel, dim3((unsigned)N), dim3(256), 0, stream, buf, 2 * (int64_t)C, gamma, nx, gx, t, C, eps);
    if (arena) iseg_deferred_push(buf, (int)N, 2 * (int64_t)C, 2 * (int64_t)C, dgamma, dbeta, C, 1.f, stream);
    else launch_reduce_rows(buf, (int)N, 2 * (int64_t)C, 0, 1, 2 * (int64_t)C, dgamma, dbeta, C, 0, 1.f, accumulate, stream);
    const int64_t chunks = N * HW * (C / 8);
    const unsigned blocks = (unsigned)(ceil_div64(chunks, 256) < 8192 ? ceil_div64(chunks, 256) : 8192);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((grn_apply_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, gamma,
                           (const float*)nullptr, nx, (const float*)t, (const bf16_t*)mul, (bf16_t*)dx, HW, C, chunks);
    else
        hipLaunchKernelGGL((grn_apply_kernel<float, true>), dim3(blocks), dim3(256), 0, stream, (const float*)x, (const float*)dy, gamma,
                           (const float*)nullptr, nx, (const float*)t, (const float*)mul, (float*)dx, HW, C, chunks);
    return iseg_check_launch("iseg_grn_bwd");
}
