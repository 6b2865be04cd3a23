// Dense contraction D[M,N] = epilogue( sum_k A(m,k) * B(k,n) ) for gfx950.
//
// Replaces every keras.layers.Dense / 1x1 Conv2D / im2col'd Conv2D matmul on the reference's hot path
// (backbones/convnext.py:29-30,51-54 pwconv1/pwconv2; layers/model_builder.py:54-64 ConvNormAct.conv;
// layers/core_model_ext.py:129 logits_conv) together with their autodiff transposes (dgrad, wgrad).
//
// Two arithmetic paths share one epilogue:
//   * bf16 storage  -> v_mfma_f32_16x16x32_bf16, fp32 accumulate (the measured path)
//   * fp32 storage  -> plain fp32 FMA tiles (the parity path; bit-comparable to a k-ordered fmaf chain)
//
// Operand orientation is described, not copied: each operand is either K-contiguous (activations [M][K],
// transposed weights [N][K]) or MN-contiguous (Keras kernels [K][N], transposed activations for wgrad).
// MN-contiguous tiles are staged in LDS as they lie in HBM (16-B coalesced loads, 16-B LDS writes) and are
// turned into MFMA fragments by ds_read_b64_tr_b16, so no transposed weight copies exist anywhere.
#include "common.h"
#include "iseg_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Epilogue (shared by the MFMA kernel, the fp32 kernel and the split-K reducer)
// ------------------------------------------------------------------------------------------------
struct Epi {
    const float* bias;
    const float* colscale;
    const float* rowscale;
    int64_t rows_per_group;
    const void* residual;
    int64_t ldr;
    const void* aux;
    int64_t ldaux;
    void* pre_out;
    int64_t ldp;
    int act;
    float alpha;
    int accumulate;
};

template <class TO>
__device__ __forceinline__ float epi_apply(const Epi& e, float acc, int64_t m, int64_t n, const TO* D, int64_t ldd) {
    float v = acc * e.alpha;
    if (e.bias) v += e.bias[n];
    if (e.pre_out) reinterpret_cast<TO*>(e.pre_out)[m * e.ldp + n] = from_f32<TO>(v);
    if (e.act == ISEG_ACT_RELU) v = fmaxf(v, 0.f);
    else if (e.act == ISEG_ACT_GELU) v = gelu_erf(v);
    else if (e.act == ISEG_ACT_GELU_GRAD) v *= gelu_erf_grad(to_f32(reinterpret_cast<const TO*>(e.aux)[m * e.ldaux + n]));
    else if (e.act == ISEG_ACT_RELU_GRAD) v = to_f32(reinterpret_cast<const TO*>(e.aux)[m * e.ldaux + n]) > 0.f ? v : 0.f;
    if (e.colscale) v *= e.colscale[n];
    if (e.rowscale) v *= e.rowscale[m / e.rows_per_group];
    if (e.residual) v += to_f32(reinterpret_cast<const TO*>(e.residual)[m * e.ldr + n]);
    if (e.accumulate) v += to_f32(D[m * ldd + n]);
    return v;
}

// 8 consecutive columns of one row, all pointers 16-B friendly (checked by the caller)
template <class TO>
__device__ __forceinline__ void epi_apply8(const Epi& e, float* v, int64_t m, int64_t n, TO* D, int64_t ldd) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
    if (e.bias) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += e.bias[n + i];
    }
    if (e.pre_out) store8<TO>(reinterpret_cast<TO*>(e.pre_out) + m * e.ldp + n, v);
    if (e.act == ISEG_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (e.act == ISEG_ACT_GELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu_erf(v[i]);
    } else if (e.act == ISEG_ACT_GELU_GRAD) {
        float a[8];
        load8<TO>(reinterpret_cast<const TO*>(e.aux) + m * e.ldaux + n, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= gelu_erf_grad(a[i]);
    } else if (e.act == ISEG_ACT_RELU_GRAD) {
        float a[8];
        load8<TO>(reinterpret_cast<const TO*>(e.aux) + m * e.ldaux + n, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = a[i] > 0.f ? v[i] : 0.f;
    }
    if (e.colscale) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= e.colscale[n + i];
    }
    if (e.rowscale) {
        const float s = e.rowscale[m / e.rows_per_group];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= s;
    }
    if (e.residual) {
        float r[8];
        load8<TO>(reinterpret_cast<const TO*>(e.residual) + m * e.ldr + n, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += r[i];
    }
    if (e.accumulate) {
        float r[8];
        load8<TO>(D + m * ldd + n, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += r[i];
    }
    store8<TO>(D + m * ldd + n, v);
}

// bijective XCD-aware remap: blocks that share an XCD (b % 8) get a contiguous run of tiles, so the
// N-tiles that re-read one A row-panel hit the same L2 (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
constexpr int BK = 32;
constexpr int KPAD = 8;   // K-contiguous tiles: row stride BK + 8 elements (80 B)
constexpr int MNPAD = 8;  // MN-contiguous tiles: row stride B{M,N} + 8 elements

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

template <int ROWS, bool KC> struct TileGeom {
    // LDS element count and row stride of one operand tile (ROWS = BM or BN)
    static constexpr int stride = KC ? (BK + KPAD) : (ROWS + MNPAD);
    static constexpr int elems = KC ? ROWS * stride : BK * stride;
    static constexpr int chunks = ROWS * BK / 8;  // 16-B chunks in the tile
};

// Load one 16-B chunk (8 bf16) of an operand tile from global memory, zero-filled out of range.
//  KC:  element (r, k) at base[r * ld + k]   (r along M or N)
//  !KC: element (r, k) at base[k * ld + r]
template <bool KC>
__device__ __forceinline__ bf16x8 load_chunk(const bf16_t* __restrict__ base, int64_t ld, int64_t r0, int64_t k0,
                                             int64_t R, int64_t Kend, bool vec) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f;
    if (KC) {
        if (r0 >= R || k0 >= Kend) return v;
        const bf16_t* p = base + r0 * ld + k0;
        if (vec && k0 + 8 <= Kend) return *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (k0 + i < Kend) v[i] = p[i];
    } else {
        if (k0 >= Kend || r0 >= R) return v;
        const bf16_t* p = base + k0 * ld + r0;
        if (vec && r0 + 8 <= R) return *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (r0 + i < R) v[i] = p[i];
    }
    return v;
}

template <int ROWS, bool KC, int NTHREADS> struct Stager {
    using G = TileGeom<ROWS, KC>;
    static constexpr int PER_THREAD = (G::chunks + NTHREADS - 1) / NTHREADS;
    bf16x8 regs[PER_THREAD];

    // chunk c -> (tile row r, tile k)
    __device__ __forceinline__ static void decode(int c, int& r, int& k) {
        if (KC) {
            r = c / (BK / 8);
            k = (c % (BK / 8)) * 8;
        } else {
            k = c / (ROWS / 8);
            r = (c % (ROWS / 8)) * 8;
        }
    }
    __device__ __forceinline__ void load(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t k0, int64_t R,
                                         int64_t Kend, bool vec, int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NTHREADS;
            if (G::chunks % NTHREADS == 0 || c < G::chunks) {
                int r, k;
                decode(c, r, k);
                regs[i] = load_chunk<KC>(base, ld, row0 + r, k0 + k, R, Kend, vec);
            }
        }
    }
    __device__ __forceinline__ void store(bf16_t* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NTHREADS;
            if (G::chunks % NTHREADS == 0 || c < G::chunks) {
                int r, k;
                decode(c, r, k);
                bf16_t* p = KC ? lds + r * G::stride + k : lds + k * G::stride + r;
                *reinterpret_cast<bf16x8*>(p) = regs[i];
            }
        }
    }
};

// 16(row) x 32(k) MFMA operand fragment from an LDS tile; `r0` = first tile row of the fragment.
template <int ROWS, bool KC>
__device__ __forceinline__ bf16x8 read_frag(const bf16_t* lds, int r0, int lane) {
    using G = TileGeom<ROWS, KC>;
    if (KC) {
        // lane holds row (lane&15), k = 8*(lane>>4) .. +7 : one ds_read_b128
        return *reinterpret_cast<const bf16x8*>(lds + (r0 + (lane & 15)) * G::stride + 8 * (lane >> 4));
    } else {
        // tile is [k][row]; ds_read_b64_tr_b16: lane 4q+p of each 16-lane group addresses LDS row q,
        // columns 4p..4p+3, and receives column (lane&15) of the four rows.
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const bf16_t* a0 = lds + (8 * g + q) * G::stride + r0 + 4 * p;
        const bf16_t* a1 = a0 + 4 * G::stride;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a1));
        bf16x8 f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[i] = lo[i];
            f[4 + i] = hi[i];
        }
        return f;
    }
}

template <int WM, int WN, int FM, int FN, bool AKC, bool BKC, class TO>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                 const bf16_t* __restrict__ B, int64_t ldb, TO* __restrict__ D,
                                                                 int64_t ldd, int64_t M, int64_t N, int64_t K, int tiles_n,
                                                                 int ntiles, int64_t k_per_split, float* __restrict__ slabs,
                                                                 Epi epi, int vecA, int vecB, int vecD) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
    constexpr int TM = FM * 16, TN = FN * 16;
    using GA = TileGeom<BM, AKC>;
    using GB = TileGeom<BN, BKC>;
    constexpr int STAGE_ELEMS = GA::elems + GB::elems;
    constexpr int EPI_ROWS = 32;  // rows of a wave tile staged per epilogue pass
    constexpr int EPI_STRIDE = TN + 4;
    constexpr int EPI_BYTES = WM * WN * EPI_ROWS * EPI_STRIDE * 4;
    constexpr int LDS_BYTES = (2 * STAGE_ELEMS * 2 > EPI_BYTES) ? 2 * STAGE_ELEMS * 2 : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    bf16_t* const lds = reinterpret_cast<bf16_t*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;

    const int t = xcd_remap(blockIdx.x, ntiles);
    const int tile_n = t % tiles_n, tile_m = t / tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int64_t kbeg = (int64_t)blockIdx.y * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    const int nk = (int)((kend - kbeg + BK - 1) / BK);

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Stager<BM, AKC, NT> sa;
    Stager<BN, BKC, NT> sb;

    if (nk > 0) {
        sa.load(A, lda, m0, kbeg, M, kend, vecA != 0, tid);
        sb.load(B, ldb, n0, kbeg, N, kend, vecB != 0, tid);
        sa.store(lds, tid);
        sb.store(lds + GA::elems, tid);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        bf16_t* cur = lds + (kt & 1) * STAGE_ELEMS;
        bf16_t* nxt = lds + ((kt + 1) & 1) * STAGE_ELEMS;
        const bool more = kt + 1 < nk;
        if (more) {
            const int64_t k0 = kbeg + (int64_t)(kt + 1) * BK;
            sa.load(A, lda, m0, k0, M, kend, vecA != 0, tid);
            sb.load(B, ldb, n0, k0, N, kend, vecB != 0, tid);
        }
        bf16x8 af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = read_frag<BM, AKC>(cur, wm * TM + i * 16, lane);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = read_frag<BN, BKC>(cur + GA::elems, wn * TN + j * 16, lane);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if (more) {
            sa.store(nxt, tid);
            sb.store(nxt + GA::elems, tid);
        }
        __syncthreads();
    }

    // ---- epilogue: accumulators -> per-wave LDS slab (fp32) -> 8-column coalesced rows ----
    float* const ew = reinterpret_cast<float*>(smem) + wid * EPI_ROWS * EPI_STRIDE;
    const bool split = slabs != nullptr;
    float* const slab = split ? slabs + (int64_t)blockIdx.y * M * N : nullptr;
    constexpr int PASSES = (TM + EPI_ROWS - 1) / EPI_ROWS;
    constexpr int FPP = EPI_ROWS / 16;  // fragments (in M) per pass
    constexpr int CPR = TN / 8;         // 8-column groups per row
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
        for (int fi = 0; fi < FPP; ++fi) {
            const int i = ps * FPP + fi;
            if (i < FM) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ew[(fi * 16 + (lane >> 4) * 4 + r) * EPI_STRIDE + j * 16 + (lane & 15)] = acc[i][j][r];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the slab is wave-private, no barrier needed
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < EPI_ROWS * CPR; c += 64) {
            const int rr = c / CPR, cc = (c % CPR) * 8;
            if (ps * EPI_ROWS + rr >= TM) continue;
            const int64_t m = m0 + wm * TM + ps * EPI_ROWS + rr;
            const int64_t n = n0 + wn * TN + cc;
            if (m >= M || n >= N) continue;
            float v[8];
            const float* src = ew + rr * EPI_STRIDE + cc;
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[u];
            if (split) {
                float* dst = slab + m * N + n;
                for (int u = 0; u < 8 && n + u < N; ++u) dst[u] = v[u];
            } else if (vecD && n + 8 <= N) {
                epi_apply8<TO>(epi, v, m, n, D, ldd);
            } else {
                for (int u = 0; u < 8 && n + u < N; ++u) D[m * ldd + n + u] = from_f32<TO>(epi_apply<TO>(epi, v[u], m, n + u, D, ldd));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 kernel (parity path): 64x64 tile, 16-deep k step, 4x4 outputs per thread, k-ordered fmaf
// ------------------------------------------------------------------------------------------------
template <class TO>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                       const float* __restrict__ B, int64_t sbk, int64_t sbn,
                                                       TO* __restrict__ D, int64_t ldd, int64_t M, int64_t N, int64_t K,
                                                       int tiles_n, int64_t k_per_split, float* __restrict__ slabs, Epi epi) {
    constexpr int TB = 64, TK = 16;
    __shared__ float sA[TK][TB + 1];
    __shared__ float sB[TK][TB + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int64_t m0 = (int64_t)tile_m * TB, n0 = (int64_t)tile_n * TB;
    const int64_t kbeg = (int64_t)blockIdx.y * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    float acc[4][4] = {};
    for (int64_t k0 = kbeg; k0 < kend; k0 += TK) {
        for (int e = tid; e < TB * TK; e += 256) {
            // pick the faster-varying index to follow the contiguous dimension of each operand
            int ka, ma;
            if (sak == 1) { ka = e % TK; ma = e / TK; } else { ma = e % TB; ka = e / TB; }
            const int64_t m = m0 + ma, k = k0 + ka;
            sA[ka][ma] = (m < M && k < kend) ? A[m * sam + k * sak] : 0.f;
            int kb, nb;
            if (sbk == 1) { kb = e % TK; nb = e / TK; } else { nb = e % TB; kb = e / TB; }
            const int64_t n = n0 + nb, k2 = k0 + kb;
            sB[kb][nb] = (n < N && k2 < kend) ? B[k2 * sbk + n * sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sB[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    const bool split = slabs != nullptr;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m >= M || n >= N) continue;
            if (split) slabs[(int64_t)blockIdx.y * M * N + m * N + n] = acc[i][j];
            else D[m * ldd + n] = from_f32<TO>(epi_apply<TO>(epi, acc[i][j], m, n, D, ldd));
        }
}

// sum split-K slabs in slab order (deterministic) and apply the epilogue
template <class TO>
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int nsplit, TO* __restrict__ D, int64_t ldd, int64_t M,
                                     int64_t N, Epi epi) {
    const int64_t total = M * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += slabs[(int64_t)z * total + i];
        const int64_t m = i / N, n = i % N;
        D[m * ldd + n] = from_f32<TO>(epi_apply<TO>(epi, s, m, n, D, ldd));
    }
}

template <int WM, int WN, int FM, int FN, bool AKC, bool BKC, class TO>
void launch_bf16(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t k_per_split, float* slabs, hipStream_t s) {
    constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
    const int tiles_m = (int)ceil_div64(g->M, BM), tiles_n = (int)ceil_div64(g->N, BN);
    const int ntiles = tiles_m * tiles_n;
    const bf16_t* A = (const bf16_t*)g->A;
    const bf16_t* B = (const bf16_t*)g->B;
    const int vecA = ((uintptr_t)A % 16 == 0) && (g->lda % 8 == 0);
    const int vecB = ((uintptr_t)B % 16 == 0) && (g->ldb % 8 == 0);
    int vecD = ((uintptr_t)g->D % 16 == 0) && (g->ldd % 8 == 0);
    if (g->residual) vecD = vecD && ((uintptr_t)g->residual % 16 == 0) && (g->ldr % 8 == 0);
    if (g->aux) vecD = vecD && ((uintptr_t)g->aux % 16 == 0) && (g->ldaux % 8 == 0);
    if (g->pre_out) vecD = vecD && ((uintptr_t)g->pre_out % 16 == 0) && (g->ldp % 8 == 0);
    dim3 grid(ntiles, nsplit);
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, FM, FN, AKC, BKC, TO>), grid, dim3(WM * WN * 64), 0, s, A, g->lda, B, g->ldb,
                       (TO*)g->D, g->ldd, g->M, g->N, g->K, tiles_n, ntiles, k_per_split, slabs, epi, vecA, vecB, vecD);
}

template <bool AKC, bool BKC, class TO>
void dispatch_tile(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    const int64_t N = g->N;
    if (N <= 32) launch_bf16<4, 1, 2, 2, AKC, BKC, TO>(g, epi, nsplit, kps, slabs, s);
    else if (N <= 64) launch_bf16<2, 2, 4, 2, AKC, BKC, TO>(g, epi, nsplit, kps, slabs, s);
    else if (N % 128 != 0 && N % 96 == 0) launch_bf16<2, 2, 4, 3, AKC, BKC, TO>(g, epi, nsplit, kps, slabs, s);
    else launch_bf16<2, 2, 4, 4, AKC, BKC, TO>(g, epi, nsplit, kps, slabs, s);
}

template <class TO>
void dispatch_orient(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    if (g->a_kcontig && g->b_kcontig) dispatch_tile<true, true, TO>(g, epi, nsplit, kps, slabs, s);
    else if (g->a_kcontig && !g->b_kcontig) dispatch_tile<true, false, TO>(g, epi, nsplit, kps, slabs, s);
    else if (!g->a_kcontig && !g->b_kcontig) dispatch_tile<false, false, TO>(g, epi, nsplit, kps, slabs, s);
    else dispatch_tile<false, true, TO>(g, epi, nsplit, kps, slabs, s);
}

// choose the split so that a skinny-output contraction (wgrad: M,N small, K = pixels) still fills 256 CUs
int choose_split(const iseg_gemm_args* g, int tile) {
    if (g->split_k > 0) return g->split_k;
    const int64_t tiles = ceil_div64(g->M, tile) * ceil_div64(g->N, tile);
    if (tiles >= 256 || g->K < 2048) return 1;
    int64_t want = ceil_div64(1024, tiles);
    const int64_t maxs = g->K / 512 > 0 ? g->K / 512 : 1;
    if (want > maxs) want = maxs;
    if (want > 512) want = 512;
    return (int)(want < 1 ? 1 : want);
}

}  // namespace

extern "C" int iseg_gemm_splits(const iseg_gemm_args* g) {
    return choose_split(g, g->in_dtype == ISEG_BF16 ? 128 : 64);
}

extern "C" size_t iseg_gemm_workspace_bytes(const iseg_gemm_args* g) {
    const int s = iseg_gemm_splits(g);
    return s > 1 ? (size_t)s * (size_t)g->M * (size_t)g->N * sizeof(float) : 0;
}

extern "C" int iseg_gemm(const iseg_gemm_args* g, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(g && g->A && g->B && g->D, "iseg_gemm: null operand");
    ISEG_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0, "iseg_gemm: empty problem M=%lld N=%lld K=%lld", (long long)g->M,
                 (long long)g->N, (long long)g->K);
    ISEG_REQUIRE(g->in_dtype == ISEG_F32 || g->in_dtype == ISEG_BF16, "iseg_gemm: bad in_dtype %d", g->in_dtype);
    ISEG_REQUIRE(g->out_dtype == ISEG_F32 || g->out_dtype == ISEG_BF16, "iseg_gemm: bad out_dtype %d", g->out_dtype);
    ISEG_REQUIRE(!(g->in_dtype == ISEG_F32 && g->out_dtype == ISEG_BF16), "iseg_gemm: f32 inputs need f32 output");
    ISEG_REQUIRE(!g->rowscale || g->rows_per_group > 0, "iseg_gemm: rowscale needs rows_per_group");
    ISEG_REQUIRE((g->act != ISEG_ACT_GELU_GRAD && g->act != ISEG_ACT_RELU_GRAD) || g->aux, "iseg_gemm: act needs aux");
    Epi epi{g->bias, g->colscale, g->rowscale, g->rows_per_group, g->residual, g->ldr, g->aux, g->ldaux, g->pre_out, g->ldp,
            g->act, g->alpha, g->accumulate};
    const int nsplit = iseg_gemm_splits(g);
    float* slabs = nullptr;
    int64_t kps = g->K;
    if (nsplit > 1) {
        const size_t need = (size_t)nsplit * g->M * g->N * sizeof(float);
        if (!ws || ws_bytes < need) {
            iseg_set_error("iseg_gemm: split-K needs %zu workspace bytes, got %zu", need, ws_bytes);
            return ISEG_ERR_WORKSPACE;
        }
        slabs = (float*)ws;
        kps = ceil_div64(ceil_div64(g->K, nsplit), BK) * BK;
    }
    const int eff_split = (int)ceil_div64(g->K, kps);
    if (g->in_dtype == ISEG_BF16) {
        if (g->out_dtype == ISEG_BF16) dispatch_orient<bf16_t>(g, epi, eff_split, kps, slabs, stream);
        else dispatch_orient<float>(g, epi, eff_split, kps, slabs, stream);
    } else {
        const int tiles_m = (int)ceil_div64(g->M, 64), tiles_n = (int)ceil_div64(g->N, 64);
        const int64_t sam = g->a_kcontig ? g->lda : 1, sak = g->a_kcontig ? 1 : g->lda;
        const int64_t sbk = g->b_kcontig ? 1 : g->ldb, sbn = g->b_kcontig ? g->ldb : 1;
        hipLaunchKernelGGL((gemm_f32_kernel<float>), dim3(tiles_m * tiles_n, eff_split), dim3(256), 0, stream,
                           (const float*)g->A, sam, sak, (const float*)g->B, sbk, sbn, (float*)g->D, g->ldd, g->M, g->N, g->K,
                           tiles_n, kps, slabs, epi);
    }
    if (slabs) {
        const int64_t total = g->M * g->N;
        const int blocks = (int)(ceil_div64(total, 256) < 2048 ? ceil_div64(total, 256) : 2048);
        if (g->out_dtype == ISEG_BF16)
            hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, slabs, eff_split,
                               (bf16_t*)g->D, g->ldd, g->M, g->N, epi);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(blocks), dim3(256), 0, stream, slabs, eff_split,
                               (float*)g->D, g->ldd, g->M, g->N, epi);
    }
    return iseg_check_launch("iseg_gemm");
}
