// LDS-DMA main loop for the weight-gradient orientation: D[M,N] = A^T B with A stored [K][M] and B stored [K][N] (both MN-contiguous: the
// activations X [pixels][C] and the output gradient dY [pixels][N] as they sit in HBM), reduction split over workgroups, fp32 slabs out.
// The register-staged kernel of gemm_impl.h serves these problems with 128 x 128 tiles (staging registers leave a wavefront a 64 x 32 tile):
// at the ConvNeXt stage-2 shape (M = 1536, N = 384, K = 16384) its L2 -> LDS fills move 302 MB for 63 MB of operands and the matrix cores are
// busy 20 % of the launch (profiles/r04_pmc.json).  Here:
//   * operand tiles go HBM -> LDS by global_load_lds_dwordx4 in their native orientation ([k][m] rows of BM x 2 bytes, one instruction = 1 KiB
//     = 2 k-rows of a 256-wide tile or 4 of a 128-wide one), a ring of NS stages of 64 k-rows with NS-1 in flight, counted s_waitcnt vmcnt and
//     ONE raw s_barrier per K-step -- the loop of gemm_dma.h;
//   * no staging registers, so a wavefront owns 64 x 64 outputs and the workgroup 256 x 128 or 128 x 256: per MFMA 512 LDS bytes instead of
//     768, per tile-MAC 0.75x the L2 -> LDS bytes of 128 x 128;
//   * fragments come out of the [k][m] image by ds_read_b64_tr_b16 (two per 16 x 32 fragment).  The 16-byte chunk index of a k-row is XOR-ed
//     with a key of the row (bits 1-3: 2 (k & 3) + 8 ((k >> 3) & 1)) on the per-lane SOURCE address of the DMA and again on the read address:
//     the four k-rows a 16-lane group reads (32 B each, a whole number of 256-B bank sweeps apart) land on four different 32-B bank groups,
//     and the two row groups of a 32-lane half on the other four;
//   * B is the first MFMA operand: lane (g, c) holds output row 16 i + c, columns 16 j + 4 g .. + 3 and stores 16-B vectors into the slab;
//   * the bias-gradient ones-row of the register kernel (output row M = column sums of B) is one more MFMA per B fragment against a constant
//     fragment of ones in the wavefronts that own the first 64 rows of the first tile row -- no extra tile row.
#pragma once
#include "gemm_dma.h"

namespace iseg_mm {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static __device__ uint4 tn_zero_page;      // sixteen zero bytes (device globals are zero-initialised): the source of reduction rows past K

__device__ __forceinline__ int tn_key(int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); }

// one (tile t, reduction split ksplit) of a problem; the kernels below only decide which problem and which tile a workgroup takes
template <int WM, int WN, int NS>
__device__ __forceinline__ void tn_tile(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int64_t M, int64_t N,
                                        int64_t K, int tiles_n, int t, int ksplit, int64_t k_per_split, float* __restrict__ slabs, int ones_row) {
    constexpr int NW = WM * WN, FM = 4, FN = 4;      // (the waits in `compute` name FM = FN = 4 fragment pairs)
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int ROWA = BM * 2, ROWB = BN * 2;                 // bytes of one k-row of the A / B image
    constexpr int CPA = BM / 8, CPB = BN / 8;                   // 16-byte chunks per k-row
    constexpr int PA = 64 * ROWA / 1024, PB = 64 * ROWB / 1024, PIECES = PA + PB, PPW = PIECES / NW;
    constexpr int STAGE = 64 * (ROWA + ROWB);
    static_assert(PA % NW == 0 && PB % NW == 0, "A and B pieces must each divide over the wavefronts");
    static_assert((CPA & (CPA - 1)) == 0 && (CPB & (CPB - 1)) == 0 && CPA >= 16 && CPB >= 16, "the chunk swizzle needs power-of-two rows of >= 16 chunks");
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // NS * STAGE bytes

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int64_t m0 = (int64_t)(t / tiles_n) * BM, n0 = (int64_t)(t % tiles_n) * BN;
    const int64_t kbeg = (int64_t)ksplit * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    const int nk = (int)((kend - kbeg + 63) / 64);      // the last stage of the last split may be ragged: its rows past K read a zero page

    // per-lane DMA sources: piece P = wid + p * NW of a stage; 16-byte slot S = 64 P + lane of the A image (P < PA) or of the B image
    const bf16_t* src[PPW];
    int left[PPW];      // reduction rows from this lane's row of the piece to the end of the split's range (<= 0: a row past K)
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int P = wid + p * NW;
        if (p < PA / NW) {      // (compile time: PA is a multiple of NW, so P < PA exactly when p < PA / NW)
            const int S = P * 64 + lane, k = S / CPA, chunk = (S % CPA) ^ tn_key(k);
            int64_t col = m0 + chunk * 8;
            col = col + 8 <= M ? col : M - 8;      // columns past M: any valid chunk (their products are never stored)
            src[p] = A + (kbeg + k) * lda + col;
            left[p] = (int)(kend - kbeg) - k;
        } else {
            const int S = (P - PA) * 64 + lane, k = S / CPB, chunk = (S % CPB) ^ tn_key(k);
            int64_t col = n0 + chunk * 8;
            col = col + 8 <= N ? col : N - 8;
            src[p] = B + (kbeg + k) * ldb + col;
            left[p] = (int)(kend - kbeg) - k;
        }
    }
    const bf16_t* const zero = reinterpret_cast<const bf16_t*>(&tn_zero_page);
    auto issue = [&](int stage) {
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            // (a select, not a branch: rows past the end of a ragged reduction -- K = 17424 pixel rows of a 513 x 513 crop at stride 16 -- contribute zeros)
            const bf16_t* const from = left[p] > 0 ? src[p] : zero;
            __builtin_amdgcn_global_load_lds((glb_void_ptr)from, (lds_void_ptr)(smem + stage * STAGE + (wid + p * NW) * 1024), 16, 0, 0);
            src[p] += p < PA / NW ? 64 * lda : 64 * ldb;
            left[p] -= 64;
        }
    };

    // fragment addresses (see read_frag of gemm_impl.h for the lane roles of ds_read_b64_tr_b16): lane (g, q, p) addresses k-row 8 g + q
    // (+ 4 for the second half, + 32 for the second k-step: neither changes the key), columns 16 f + 4 p .. + 3 of the wave tile
    const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3, c15 = lane & 15;
    const int key = (q << 1) | ((g & 1) << 3);
    int a_off[FM], b_off[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) a_off[i] = (8 * g + q) * ROWA + ((((wm * FM + i) * 2) ^ key) + (p4 >> 1)) * 16 + (p4 & 1) * 8;
#pragma unroll
    for (int j = 0; j < FN; ++j) b_off[j] = 64 * ROWA + (8 * g + q) * ROWB + ((((wn * FN + j) * 2) ^ key) + (p4 >> 1)) * 16 + (p4 & 1) * 8;

    f32x4 acc[FM][FN], acc1[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < FN; ++j) acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool ones = ones_row && m0 == 0 && wm == 0;      // wavefront-uniform
    bf16x8 one8;
#pragma unroll
    for (int u = 0; u < 8; ++u) one8[u] = (bf16_t)1.0f;

    // Fragment reads are inline assembly: behind the ds_read_b64_tr_b16 BUILTIN hipcc puts s_waitcnt vmcnt(0) (it carries no alias information
    // against the LDS-DMA in flight), which drains the ring on every K-step.  The assembly form is invisible to hipcc's wait-count pass, so the
    // waits are written out too; they name the registers they guard as in/out operands, which keeps the MFMAs behind them.
    // LDS returns data in order: after [16 reads of k-step 0][8 A reads of k-step 1], lgkmcnt(8) says k-step 0 is in registers.
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_ptr)smem;
    auto compute = [&](int stage) {
        const unsigned sbase = lds0 + stage * STAGE;
        u32x2 ra[2][FM][2], rb[2][FN][2];
        unsigned aa[FM], ab[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) aa[i] = sbase + a_off[i];
#pragma unroll
        for (int j = 0; j < FN; ++j) ab[j] = sbase + b_off[j];
        // (ISEG_TN_ABL_*: ablation builds of tools/micro/tn_bench.hip -- results are wrong, timings tell what the loop is made of)
#ifdef ISEG_TN_ABL_NOREAD
#define ISEG_TR_READ(dst, addr, OFF) asm volatile("" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#else
#define ISEG_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#endif
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            ISEG_TR_READ(ra[0][i][0], aa[i], 0);
            ISEG_TR_READ(ra[0][i][1], aa[i], 4 * ROWA);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            ISEG_TR_READ(rb[0][j][0], ab[j], 0);
            ISEG_TR_READ(rb[0][j][1], ab[j], 4 * ROWB);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            ISEG_TR_READ(ra[1][i][0], aa[i], 32 * ROWA);
            ISEG_TR_READ(ra[1][i][1], aa[i], 32 * ROWA + 4 * ROWA);
        }
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(ra[0][0][0]), "+v"(ra[0][0][1]), "+v"(ra[0][1][0]), "+v"(ra[0][1][1]), "+v"(ra[0][2][0]), "+v"(ra[0][2][1]),
                       "+v"(ra[0][3][0]), "+v"(ra[0][3][1]), "+v"(rb[0][0][0]), "+v"(rb[0][0][1]), "+v"(rb[0][1][0]), "+v"(rb[0][1][1]),
                       "+v"(rb[0][2][0]), "+v"(rb[0][2][1]), "+v"(rb[0][3][0]), "+v"(rb[0][3][1])
                     :
                     : "memory");
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            ISEG_TR_READ(rb[1][j][0], ab[j], 32 * ROWB);
            ISEG_TR_READ(rb[1][j][1], ab[j], 32 * ROWB + 4 * ROWB);
        }
#undef ISEG_TR_READ
        auto join = [](const u32x2& lo, const u32x2& hi) {
            const u32x4 v{lo.x, lo.y, hi.x, hi.y};
            return __builtin_bit_cast(bf16x8, v);
        };
        auto mfmas = [&](int ks) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = join(ra[ks][i][0], ra[ks][i][1]);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = join(rb[ks][j][0], rb[ks][j][1]);
#ifdef ISEG_TN_ABL_NOMFMA
            acc[0][0][0] += __builtin_bit_cast(float, ra[ks][0][0].x ^ rb[ks][0][0].x);
#else
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
#endif
            if (ones) {
#pragma unroll
                for (int j = 0; j < FN; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], one8, acc1[j], 0, 0, 0);
            }
        };
        mfmas(0);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ra[1][0][0]), "+v"(ra[1][0][1]), "+v"(ra[1][1][0]), "+v"(ra[1][1][1]), "+v"(ra[1][2][0]), "+v"(ra[1][2][1]),
                       "+v"(ra[1][3][0]), "+v"(ra[1][3][1]), "+v"(rb[1][0][0]), "+v"(rb[1][0][1]), "+v"(rb[1][1][0]), "+v"(rb[1][1][1]),
                       "+v"(rb[1][2][0]), "+v"(rb[1][2][1]), "+v"(rb[1][3][0]), "+v"(rb[1][3][1])
                     :
                     : "memory");
        mfmas(1);
    };

    // ring of NS stages, NS-1 tiles in flight, one barrier per K-step (the loop of gemm_bf16_dma_kernel)
    static_assert(NS >= 3, "the one-barrier ring needs three stages");
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s);
    int stage = 0, fill = (NS - 1) % NS;
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt;
#ifndef ISEG_TN_ABL_NOWAIT
        if (ahead >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
        else if (NS > 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifndef ISEG_TN_ABL_NODMA
        if (kt + NS - 1 < nk) issue(fill);
#endif
        compute(stage);
        stage = stage + 1 == NS ? 0 : stage + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
    }

    // ---- slab stores straight from the accumulators: lane (g, c) holds row 16 i + c, columns 16 j + 4 g .. + 3 ----
    const int64_t slab_rows = M + (ones_row ? 1 : 0);
    float* const slab = slabs + (int64_t)ksplit * slab_rows * N;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + c15;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int64_t n = n0 + wn * 64 + j * 16 + 4 * g;
            if (m < M && n < N) *reinterpret_cast<float4*>(slab + m * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
    if (ones && c15 == 0) {      // every row of acc1 is the column sum; row 0's lanes write the ones-row (slab row M)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int64_t n = n0 + wn * 64 + j * 16 + 4 * g;
            if (n < N) *reinterpret_cast<float4*>(slab + M * N + n) = make_float4(acc1[j][0], acc1[j][1], acc1[j][2], acc1[j][3]);
        }
    }
}

template <int WM, int WN, int NS>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_dma_tn_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                                        int64_t ldb, int64_t M, int64_t N, int64_t K, int tiles_n, int ntiles,
                                                                        int64_t k_per_split, float* __restrict__ slabs, int ones_row) {
    int t, ksplit;
    tile_and_split(ntiles, t, ksplit);
    tn_tile<WM, WN, NS>(A, lda, B, ldb, M, N, K, tiles_n, t, ksplit, k_per_split, slabs, ones_row);
}

// TWO weight-gradient problems over the SAME reduction rows in one launch (round 5): the pair of an un-fused ConvNeXt block, Z = g^T dout
// [4C, C] and dW1 = y2^T dH [C, 4C] (backbones/convnext.py:51-54 backward).  Launched one after the other each fills the chip with
// 18 tiles x 13 splits; together 36 tiles x 7 splits do -- half the slab bytes written and summed again, one ring fill and one slab-store
// tail instead of two.  A workgroup takes a tile of problem 0 or 1 (its own tile form: 256 x 128 or 128 x 256 per problem).
struct TnProblem {
    const bf16_t* A;
    int64_t lda;
    const bf16_t* B;
    int64_t ldb;
    int64_t M, N;
    float* slabs;
    int tiles_n, ntiles, form, ones_row;      // form 7 = 256 x 128 tiles, 8 = 128 x 256
};

template <int NS>
__global__ __launch_bounds__(512) void gemm_bf16_dma_tn_pair_kernel(TnProblem p0, TnProblem p1, int64_t K, int64_t k_per_split) {
    int t, ksplit;
    tile_and_split(p0.ntiles + p1.ntiles, t, ksplit);
    const bool second = t >= p0.ntiles;      // workgroup-uniform
    const TnProblem& p = second ? p1 : p0;
    if (second) t -= p0.ntiles;
    if (p.form == 7) tn_tile<4, 2, NS>(p.A, p.lda, p.B, p.ldb, p.M, p.N, K, p.tiles_n, t, ksplit, k_per_split, p.slabs, p.ones_row);
    else tn_tile<2, 4, NS>(p.A, p.lda, p.B, p.ldb, p.M, p.N, K, p.tiles_n, t, ksplit, k_per_split, p.slabs, p.ones_row);
}

int dma_tn_mode();      // ISEG_GEMM_DMA_TN: 0 = never, 1 = whenever eligible (default)
bool dma_tn_lds_ok();   // gemm_tn.hip: the 144-KiB dynamic-LDS limit of both tile forms was raised (once per process); false -> the register kernel keeps these problems

// 0 = not eligible; 7 = 256 x 128 tiles, 8 = 128 x 256 tiles (the codes iseg_gemm_variant reports)
inline int dma_tn_form(const iseg_gemm_args* g) {
    if (!dma_tn_mode() || g->in_dtype != ISEG_BF16 || g->a_kcontig || g->b_kcontig || g->a_act != ISEG_ACT_NONE) return 0;
    if (!dma_tn_lds_ok()) return 0;
    if (g->batch > 1 || g->b_group_rows > 0 || g->split_k == 1) return 0;
    static const int min_mn = [] {      // ISEG_GEMM_DMA_TN_MIN: narrowest M / N the kernel takes (columns past M / N are clamped duplicates)
        const char* e = getenv("ISEG_GEMM_DMA_TN_MIN");
        const int v = e ? atoi(e) : 64;
        return v < 64 ? 64 : v;      // (the ones-row lives in the first 64 rows' wavefronts)
    }();
    if (g->M < min_mn || g->N < min_mn || g->M % 8 || g->N % 8 || g->K < 2048) return 0;      // (any K: a ragged last stage reads zeros)
    // a side below 128 pays for clamped duplicate columns: measured (tools/kbench_wgrad_tn.py, KBENCH_NARROW=1, register kernel -> this one, us)
    // 96 x 384 55.0 -> 47.4, 384 x 96 58.0 -> 45.9, 112 x 336 29.9 -> 26.0, 112 x 448 32.9 -> 27.9, but 96 x 288 46.4 -> 51.7 and 96 x 96 23.7 -> 24.4
    if ((g->M < 128 || g->N < 128) && (g->M > g->N ? g->M : g->N) < 320) return 0;
    if (((uintptr_t)g->A % 16) || ((uintptr_t)g->B % 16) || g->lda % 8 || g->ldb % 8) return 0;
    const int64_t t7 = ceil_div64(g->M, 256) * ceil_div64(g->N, 128), t8 = ceil_div64(g->M, 128) * ceil_div64(g->N, 256);
    return t8 < t7 ? 8 : 7;
}

// K splits of an eligible problem: one resident round of workgroups (one per CU: the ring takes 144 KiB), at least 512 reduction rows each
inline int dma_tn_split(const iseg_gemm_args* g, int form) {
    if (g->split_k > 1) return g->split_k;
    const int64_t tiles = form == 8 ? ceil_div64(g->M, 128) * ceil_div64(g->N, 256) : ceil_div64(g->M, 256) * ceil_div64(g->N, 128);
    int64_t want = dma_cus() / tiles;
    const int64_t maxs = g->K / 512;
    if (want > maxs) want = maxs;
    return (int)(want < 2 ? 0 : want);      // 0: not worth splitting -> the register kernel's plan
}

template <int WM, int WN>
void launch_dma_tn(const iseg_gemm_args* g, int nsplit, int64_t k_per_split, float* slabs, hipStream_t s) {
    constexpr int NS = 3, BM = WM * 64, BN = WN * 64;
    const int tiles_m = (int)ceil_div64(g->M, BM), tiles_n = (int)ceil_div64(g->N, BN);
    const int ntiles = tiles_m * tiles_n;
    constexpr int lds = NS * 64 * (BM + BN) * 2;      // (the limit was raised by dma_tn_lds_ok(), which dma_tn_form() requires)
    hipLaunchKernelGGL((gemm_bf16_dma_tn_kernel<WM, WN, NS>), dim3(ntiles, nsplit, 1), dim3(WM * WN * 64), lds, s, (const bf16_t*)g->A, g->lda,
                       (const bf16_t*)g->B, g->ldb, g->M, g->N, g->K, tiles_n, ntiles, k_per_split, slabs, g->colsum_out ? 1 : 0);
}

// reduction splits of a PAIR launch: one resident round of workgroups over both problems' tiles, at least 512 reduction rows each; 0 = do not pair
inline int dma_tn_pair_split(const iseg_gemm_args* g0, const iseg_gemm_args* g1) {
    const int f0 = dma_tn_form(g0), f1 = dma_tn_form(g1);
    if (!f0 || !f1 || g0->K != g1->K) return 0;
    auto tiles = [](const iseg_gemm_args* g, int form) {
        return form == 8 ? ceil_div64(g->M, 128) * ceil_div64(g->N, 256) : ceil_div64(g->M, 256) * ceil_div64(g->N, 128);
    };
    const int64_t t = tiles(g0, f0) + tiles(g1, f1);
    int64_t want = dma_cus() / t;
    const int64_t maxs = g0->K / 512;
    if (want > maxs) want = maxs;
    return (int)(want < 2 ? 0 : want);
}

inline void launch_dma_tn_pair(const iseg_gemm_args* g0, float* slabs0, const iseg_gemm_args* g1, float* slabs1, int nsplit, int64_t k_per_split,
                               hipStream_t s) {
    auto prob = [](const iseg_gemm_args* g, float* slabs) {
        TnProblem p;
        p.form = dma_tn_form(g);
        p.A = (const bf16_t*)g->A;
        p.lda = g->lda;
        p.B = (const bf16_t*)g->B;
        p.ldb = g->ldb;
        p.M = g->M;
        p.N = g->N;
        p.slabs = slabs;
        p.tiles_n = (int)ceil_div64(g->N, p.form == 8 ? 256 : 128);
        p.ntiles = (int)ceil_div64(g->M, p.form == 8 ? 128 : 256) * p.tiles_n;
        p.ones_row = g->colsum_out ? 1 : 0;
        return p;
    };
    const TnProblem p0 = prob(g0, slabs0), p1 = prob(g1, slabs1);
    constexpr int lds = 3 * 64 * (256 + 128) * 2;
    hipLaunchKernelGGL((gemm_bf16_dma_tn_pair_kernel<3>), dim3(p0.ntiles + p1.ntiles, nsplit, 1), dim3(512), lds, s, p0, p1, g0->K, k_per_split);
}

}  // namespace iseg_mm
