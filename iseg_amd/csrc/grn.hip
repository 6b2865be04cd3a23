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

// ---- the normalisation folded into the Dense that follows it (backbones/convnext_v2.py:92-93: x = grn(x); x = pwconv2(x)) ----------------------
//   z = g*a_n + beta with a_n = gamma*nx_n + 1 per sample, so  z W + b = g (diag(a_n) W) + (b + beta W): N scaled copies of the small kernel and
//   one bias vector replace the pass that would write z;  dW = sum_n diag(a_n) G_n + beta (x) S  with G_n = g_n^T dbr_n (per-sample weight-gradient
//   slabs) and S = colsum(dbr);  the statistics of the GRN backward come from the same slabs: sum_hw dz*g = rowsum(G_n . W), sum dz = W S.

// out[n][o][k] = wt[o][k] * (gamma[k]*nx[n][k] + 1): K-contiguous per-sample kernels for the LDS-DMA GEMM (B per row group)
__global__ __launch_bounds__(256) void grn_fold_weights_kernel(const bf16_t* __restrict__ wt, const float* __restrict__ gamma,
                                                               const float* __restrict__ nx, bf16_t* __restrict__ out, int Cout, int C4,
                                                               int64_t chunks) {
    const int nch = C4 / 8;
    const int64_t per = (int64_t)Cout * nch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / per);
        const int64_t r = i - (int64_t)n * per;
        const int k = (int)(r % nch) * 8;
        float w[8], g[8], a[8];
        load8<bf16_t>(wt + r * 8, w);
        load8<float>(gamma + k, g);
        load8<float>(nx + (int64_t)n * C4 + k, a);
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] *= fmaf(g[u], a[u], 1.f);
        store8<bf16_t>(out + i * 8, w);
    }
}

// out[c] = b[c] + sum_k beta[k] * W[k][c]      (W [C4][Cout] fp32 master; 64 columns x 16 row lanes per workgroup, four independent loads per
// lane and trip, fixed order: with 4 row lanes and one load per trip the 2-3 workgroups of a stage-0 kernel needed 58 us of pure latency)
__global__ __launch_bounds__(1024) void grn_fold_bias_kernel(const float* __restrict__ W, const float* __restrict__ beta, const float* __restrict__ b,
                                                             float* __restrict__ out, int C4, int Cout) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < Cout) {
        int k = ty;
        for (; k + 48 < C4; k += 64) {
            s0 = fmaf(beta[k], W[(int64_t)k * Cout + c], s0);
            s1 = fmaf(beta[k + 16], W[(int64_t)(k + 16) * Cout + c], s1);
            s2 = fmaf(beta[k + 32], W[(int64_t)(k + 32) * Cout + c], s2);
            s3 = fmaf(beta[k + 48], W[(int64_t)(k + 48) * Cout + c], s3);
        }
        for (; k < C4; k += 16) s0 = fmaf(beta[k], W[(int64_t)k * Cout + c], s0);
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && c < Cout) {
        float t = red[0][tx];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += red[q][tx];
        out[c] = (b ? b[c] : 0.f) + t;
    }
}

// one workgroup per kernel row k.  slabs [nslab][C4][Cout] hold G = g^T dbr of consecutive row chunks, `sps` chunks per sample.
//   dW[k][c] (+)= sum_n a_n[k] * G_n[k][c] + beta[k]*S[c];   dstats[n][k] = sum_c G_n[k][c]*W[k][c];   dstats[0][C4+k] = sum_c W[k][c]*S[c]
template <int CPT>
__global__ __launch_bounds__(256) void grn_fold_wgrad_kernel(const float* __restrict__ slabs, int sps, const float* __restrict__ W,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ nx, const float* __restrict__ S, float* __restrict__ dW,
                                                             float* __restrict__ dstats, int N, int C4, int Cout, int accumulate) {
    __shared__ float red[4];
    const int k = blockIdx.x;
    const int64_t row = (int64_t)k * Cout;
    const int64_t slab = (int64_t)C4 * Cout;
    float w[CPT], acc[CPT], sv[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = threadIdx.x + j * 256;
        w[j] = c < Cout ? W[row + c] : 0.f;
        sv[j] = c < Cout ? S[c] : 0.f;
        acc[j] = 0.f;
    }
    const float gk = gamma[k];
    for (int n = 0; n < N; ++n) {
        const float a = fmaf(gk, nx[(int64_t)n * C4 + k], 1.f);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = threadIdx.x + j * 256;
            if (c < Cout) {
                float gs = 0.f;
                for (int q = 0; q < sps; ++q) gs += slabs[((int64_t)n * sps + q) * slab + row + c];
                acc[j] = fmaf(a, gs, acc[j]);
                d = fmaf(gs, w[j], d);
            }
        }
        d = block_sum_256(d, red);
        if (threadIdx.x == 0) dstats[(int64_t)n * 2 * C4 + k] = d;
    }
    float ws = 0.f;
    const float bk = beta[k];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < Cout) {
            const float v = fmaf(bk, sv[j], acc[j]);
            dW[row + c] = accumulate ? dW[row + c] + v : v;
            ws = fmaf(w[j], sv[j], ws);
        }
    }
    ws = block_sum_256(ws, red);
    if (threadIdx.x == 0) {
        dstats[C4 + k] = ws;
        for (int n = 1; n < N; ++n) dstats[(int64_t)n * 2 * C4 + C4 + k] = 0.f;
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
    ISEG_REQUIRE(x && nx && gx && N > 0 && HW > 0 && C > 0 && (!y || (gamma && beta)), "iseg_grn_fwd: bad arguments");
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
    if (!y) return iseg_check_launch("iseg_grn_fwd");      // statistics only: the caller folds gamma*nx + 1 into the next layer's kernel
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

extern "C" int iseg_grn_fold_weights(const void* wt, const float* gamma, const float* nx, void* out, int64_t N, int Cout, int C4,
                                     hipStream_t stream) {
    ISEG_REQUIRE(wt && gamma && nx && out && N > 0 && Cout > 0 && C4 > 0 && C4 % 8 == 0, "iseg_grn_fold_weights: bad arguments");
    const int64_t chunks = N * Cout * (C4 / 8);
    const unsigned blocks = (unsigned)(ceil_div64(chunks, 256) < 4096 ? ceil_div64(chunks, 256) : 4096);
    hipLaunchKernelGGL(grn_fold_weights_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)wt, gamma, nx, (bf16_t*)out, Cout, C4, chunks);
    return iseg_check_launch("iseg_grn_fold_weights");
}

extern "C" int iseg_grn_fold_bias(const float* W, const float* beta, const float* b, float* out, int C4, int Cout, hipStream_t stream) {
    ISEG_REQUIRE(W && beta && out && C4 > 0 && Cout > 0, "iseg_grn_fold_bias: bad arguments");
    hipLaunchKernelGGL(grn_fold_bias_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(1024), 0, stream, W, beta, b, out, C4, Cout);
    return iseg_check_launch("iseg_grn_fold_bias");
}

extern "C" int iseg_grn_fold_wgrad(const float* slabs, int slabs_per_sample, const float* W, const float* gamma, const float* beta, const float* nx,
                                   const float* S, float* dW, float* dstats, int accumulate, int64_t N, int C4, int Cout, hipStream_t stream) {
    ISEG_REQUIRE(slabs && W && gamma && beta && nx && S && dW && dstats && slabs_per_sample > 0 && N > 0 && C4 > 0 && Cout > 0,
                 "iseg_grn_fold_wgrad: bad arguments");
    ISEG_REQUIRE(Cout <= 4096 && N <= 65535, "iseg_grn_fold_wgrad: at most 4096 output channels");
#define FOLD_WG(CPT_)                                                                                                                        \
    hipLaunchKernelGGL((grn_fold_wgrad_kernel<CPT_>), dim3((unsigned)C4), dim3(256), 0, stream, slabs, slabs_per_sample, W, gamma, beta, nx, S, dW, \
                       dstats, (int)N, C4, Cout, accumulate)
    if (Cout <= 256) FOLD_WG(1);
    else if (Cout <= 512) FOLD_WG(2);
    else if (Cout <= 1024) FOLD_WG(4);
    else if (Cout <= 2048) FOLD_WG(8);
    else FOLD_WG(16);
#undef FOLD_WG
    return iseg_check_launch("iseg_grn_fold_wgrad");
}

extern "C" int iseg_grn_bwd_folded(const void* dy, const void* x, const float* gamma, const float* nx, const float* gx, const void* mul,
                                   float* dstats, void* dx, float* dgamma, float* dbeta, int accumulate, int64_t N, int64_t HW, int C, float eps,
                                   int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(dy && x && gamma && nx && gx && dstats && dx && dgamma && dbeta && N > 0 && HW > 0 && C > 0, "iseg_grn_bwd_folded: bad arguments");
    ISEG_REQUIRE((dtype == ISEG_BF16 || dtype == ISEG_F32) && C % 8 == 0, "iseg_grn_bwd_folded: C %% 8 == 0 required (got dtype %d, C %d)", dtype, C);
    ISEG_REQUIRE(N <= 65535, "iseg_grn_bwd_folded: at most 65535 samples");
    ISEG_REQUIRE(ws && ws_bytes >= iseg_grn_workspace_bytes(N, HW, C), "iseg_grn_bwd_folded: workspace too small");
    float* t = (float*)ws;
    hipLaunchKernelGGL(grn_bwd_stats_kernel, dim3((unsigned)N), dim3(256), 0, stream, dstats, 2 * (int64_t)C, gamma, nx, gx, t, C, eps);
    launch_reduce_rows(dstats, (int)N, 2 * (int64_t)C, 0, 1, 2 * (int64_t)C, dgamma, dbeta, C, 0, 1.f, accumulate, stream);
    const int64_t chunks = N * HW * (C / 8);
    const unsigned blocks = (unsigned)(ceil_div64(chunks, 256) < 8192 ? ceil_div64(chunks, 256) : 8192);
    if (dtype == ISEG_BF16)
        hipLaunchKernelGGL((grn_apply_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, gamma,
                           (const float*)nullptr, nx, (const float*)t, (const bf16_t*)mul, (bf16_t*)dx, HW, C, chunks);
    else
        hipLaunchKernelGGL((grn_apply_kernel<float, true>), dim3(blocks), dim3(256), 0, stream, (const float*)x, (const float*)dy, gamma,
                           (const float*)nullptr, nx, (const float*)t, (const float*)mul, (float*)dx, HW, C, chunks);
    return iseg_check_launch("iseg_grn_bwd_folded");
}
