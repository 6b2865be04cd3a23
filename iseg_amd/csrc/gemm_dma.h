// Second bf16 main loop: LDS-DMA pipeline (global_load_lds_dwordx4) for problems whose A and B are both K-contiguous and 16-byte
// aligned, with the reduction range a multiple of 64, N a multiple of 8 and no operand transform (dgrads, attention products).
// Differences from gemm_bf16_kernel (gemm_impl.h):
//   * operand tiles go HBM -> LDS without passing through registers: no staging VGPRs, no ds_write pass, so a wavefront owns a
//     64x64 output tile (16 accumulator fragments; 8 ds_read_b128 per 16 MFMAs = 512 LDS bytes per MFMA instead of 768);
//   * a ring of NS LDS stages: with NS = 2 tile t+1's DMA is issued before tile t is waited for (two barriers per K-step, two
//     workgroups per CU); with NS >= 3, NS-1 tiles are in flight and ONE barrier per K-step does both jobs.  Waits are COUNTED
//     s_waitcnt vmcnt and the barriers raw s_barrier (a __syncthreads() would drain the DMA in flight);
//   * the LDS image of a tile is unpadded [row][64] bf16 (128-B rows; one DMA instruction = 8 rows = 1 KiB, lane-linear) with
//     the 16-B chunk index XOR-ed by a 3-bit key of the row: applied on the per-lane SOURCE address and again on the ds_read_b128
//     address, it makes every 16-lane read group ({0-3,12-15,20-27}, ...) cover all 64 banks;
//   * B is the first MFMA operand and its fragment rows are permuted, so the accumulator holds the TRANSPOSED product with eight
//     consecutive output columns per lane: the fused epilogue (epi_finish8 of gemm_impl.h: same operations, split-K slabs and
//     strided batch) runs on 16-byte vectors straight from registers -- no LDS transpose, no barrier after the main loop;
//   * rows past M / N are clamped to the last valid row on the source side (their products are never stored).
// Measured (tools/kbench_gemm_ref.py, plain epilogue): dgrad M=16384 N=1536 K=384 43.5 -> 35.2 us, M=16384 N=384 K=1536
// 44.6 -> 32.1 us, M=4096 N=4096 K=1024 59.9 -> 43.0 us; end to end the flagship gains 0.5 % (its dgrads are epilogue-bound).
#pragma once
#include "gemm_impl.h"
#include <stdlib.h>

namespace iseg_mm {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// swizzle key of a B-tile row (see the fragment read below: a 16-lane read group holds rows {x, 8+x, 16+x, 24+x} + const, x = 0..3)
__device__ __forceinline__ int b_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// EK: epilogue kind known at compile time (0 = whatever the Epi struct says at run time).  The flagship's three fused epilogues get their
// own instantiation: dead branches of epi_finish8 fold away, and a kernel trace / PMC pass can tell the forward product (gelu + gelu'
// outputs) from the data gradient (x aux) -- with one symbol for both, rocprof's per-kernel traffic was a mean over two different epilogues.
enum { EK_ANY = 0, EK_GELU_DERIV = 1, EK_MUL_AUX = 2, EK_BIAS_RESIDUAL = 3, EK_PLAIN = 4, EK_BIAS = 5 };

template <int EK> __device__ __forceinline__ Epi epi_known(Epi e) {
    if (EK == EK_GELU_DERIV) {          // bias -> gelu, second output gelu'(pre): pwconv1 forward of an un-fused ConvNeXt block
        e.act = ISEG_ACT_GELU;
        e.pre_deriv = 1;
        e.residual = nullptr;
        e.aux = nullptr;
        e.colscale = nullptr;
        e.accumulate = 0;
        e.alpha = 1.f;
        e.bias_rowscaled = 0;      // (rowscale stays a run-time option: round 6 writes rowscale * gelu(h), the drop-path factor of the block)
    } else if (EK == EK_MUL_AUX) {      // x aux (saved gelu'): pwconv2 data gradient
        e.act = ISEG_ACT_MUL_AUX;
        e.bias = nullptr;
        e.pre_out = nullptr;
        e.residual = nullptr;
        e.colscale = nullptr;
        e.accumulate = 0;
        e.alpha = 1.f;
        e.bias_rowscaled = 0;      // (rowscale stays a run-time option: the drop-path factor on the arriving gradient)
    } else if (EK == EK_PLAIN) {        // nothing fused: pwconv1 data gradient
        e.act = ISEG_ACT_NONE;
        e.bias = nullptr;
        e.aux = nullptr;
        e.pre_out = nullptr;
        e.residual = nullptr;
        e.colscale = nullptr;
        e.rowscale = nullptr;
        e.bias_rowscaled = 0;
        e.accumulate = 0;
        e.alpha = 1.f;
    } else if (EK == EK_BIAS) {         // + bias only: the Dense / projection forward products of the transformer and DCNv3 layers
        e.act = ISEG_ACT_NONE;
        e.aux = nullptr;
        e.pre_out = nullptr;
        e.residual = nullptr;
        e.colscale = nullptr;
        e.rowscale = nullptr;
        e.bias_rowscaled = 0;
        e.accumulate = 0;
        e.alpha = 1.f;
    } else if (EK == EK_BIAS_RESIDUAL) {      // bias, layer scale / drop-path factor (run time), + residual: pwconv2 forward
        e.act = ISEG_ACT_NONE;
        e.aux = nullptr;
        e.pre_out = nullptr;
        e.accumulate = 0;
        e.alpha = 1.f;
    }
    return e;
}

// sixteen zero bytes in device memory: what a lane past the K tail asks the DMA for
static __device__ __attribute__((aligned(16))) unsigned int dma_zero_chunk[4] = {0u, 0u, 0u, 0u};

int dma_min_k();      // ISEG_GEMM_DMA_MIN_K (default 32): shortest reduction the pipeline takes (262144 x 96 x 48: 22.4 -> 19.4 us)

inline int epi_kind(const Epi& e, const float* slabs) {
    if (slabs || e.alpha != 1.f || e.accumulate) return EK_ANY;
    if (e.act == ISEG_ACT_GELU && e.pre_out && e.pre_deriv && e.bias && !e.residual && !e.aux && !e.colscale && !e.bias_rowscaled) return EK_GELU_DERIV;
    if (e.act == ISEG_ACT_MUL_AUX && e.aux && !e.bias && !e.pre_out && !e.residual && !e.colscale) return EK_MUL_AUX;
    if (e.act == ISEG_ACT_NONE && e.bias && e.residual && !e.aux && !e.pre_out) return EK_BIAS_RESIDUAL;
    static const bool bias_kind = [] { const char* v = getenv("ISEG_GEMM_EK_BIAS"); return !v || atoi(v) != 0; }();      // 0: A/B against the run-time epilogue
    if (bias_kind && e.act == ISEG_ACT_NONE && e.bias && !e.residual && !e.aux && !e.pre_out && !e.colscale && !e.rowscale) return EK_BIAS;
    if (e.act == ISEG_ACT_NONE && !e.bias && !e.residual && !e.aux && !e.pre_out && !e.colscale && !e.rowscale) return EK_PLAIN;
    return EK_ANY;
}

// KT: the instantiation that handles a K tail (launched only for K % 64 != 0: the tail's address selects cost the whole-K flagship launches 0.5 %
// of the step when they were compiled into every instantiation -- tools/ab_ktail.sh, 8.28 vs 8.24 ms)
// BKS: K elements per ring stage.  64 (one 128-byte row per stage row: the form of rounds 1-5) or 32 (round 6: 64-byte stage rows, half the ring,
// so TWO 256 x 128 workgroups fit a CU -- 72 KB each -- and one's register epilogue (GELU evaluations, operand loads, stores) overlaps the other's
// main loop; one barrier per 16 MFMAs instead of per 32).  The 64-byte image is swizzled with a 2-bit key per group of four rows,
// key = {0, 3, 2, 1}[(row >> 2) & 3] for A and {0, 3, 2, 1}[(row >> 3) & 3] for the permuted B rows: every 16-lane ds_read_b128 group
// ({0-3, 12-15, 20-27}, ...) then covers the sixteen 16-byte slots of a 256-byte bank row exactly once.
__device__ __forceinline__ int key32(int quad) { return (0x6C >> (2 * (quad & 3))) & 3; }      // {0, 3, 2, 1}

// FM: 16-row blocks per wavefront (4; 2 for the 128 x 192 tile of round 6: eight wavefronts of 32 x 96)
template <int WM, int WN, int NS, class TO, bool PERSIST = false, int FN = 4, int EK = EK_ANY, bool KT = false, int BKS = 64, int FM = 4>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(BKS == 32 ? 4 : 1, BKS == 32 ? 4 : 8))) void gemm_bf16_dma_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                                     int64_t ldb, TO* __restrict__ D, int64_t ldd, int64_t M, int64_t N,
                                                                     int64_t K, int tiles_n, int ntiles, int64_t k_per_split,
                                                                     float* __restrict__ slabs, Epi epi, int vecD) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * FM * 16, BN = WN * FN * 16;      // a wavefront owns FM*16 rows x FN*16 columns (FN = 4, or 6 for the 256 x 192 tile)
    static_assert(FN % 2 == 0, "column fragments come in pairs (eight consecutive columns per lane)");
    static_assert(BKS == 64 || (BKS == 32 && !KT && !PERSIST), "ring stages hold 64 or 32 K elements; the 32 form has no K tail and no persistent walk");
    constexpr int RB = BKS * 2;                // bytes per stage row
    constexpr int CPR = BKS / 8;               // 16-byte chunks per stage row
    constexpr int RPP = 1024 / RB;             // stage rows per 1-KiB DMA piece
    constexpr int PIECES = (BM + BN) / RPP;    // 1-KiB DMA pieces per stage
    constexpr int PPW = PIECES / NW;           // pieces each wavefront issues per stage
    static_assert(PIECES % NW == 0, "stage pieces must divide over the wavefronts");
    constexpr int STAGE = (BM + BN) * RB;      // bytes
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // NS * STAGE bytes

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    if (gridDim.z > 1) {
        A += epi.off_a(blockIdx.z);
        B += epi.off_b(blockIdx.z);
        D += epi.off_d(blockIdx.z);
    }
    // PERSIST (one K split, no batch, NS >= 3): gridDim.x <= the number of CUs and a workgroup walks the virtual workgroup ids
    // blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x % 8 == 0, so every id of a workgroup maps to the same XCD and xcd_remap hands
    // it tiles of that XCD's contiguous run).  The first NS-1 stages of tile i+1 are requested BEFORE tile i's epilogue: the ring is
    // idle there (the epilogue runs from registers), so the HBM -> LDS latency of the next tile and the epilogue's own operand reads,
    // GELU evaluations and stores overlap instead of queueing.
    int t, ksplit;
    if (PERSIST) {
        t = xcd_remap(blockIdx.x, ntiles);
        ksplit = 0;
    } else {
        tile_and_split(ntiles, t, ksplit);
    }
    int64_t m0 = (int64_t)(t / tiles_n) * BM, n0 = (int64_t)(t % tiles_n) * BN;
    const int64_t kbeg = (int64_t)ksplit * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    // K tail: a last K-step of fewer than eight 16-B chunks.  The lanes whose source chunk lies beyond it request a 16-B run of zeros instead
    // (the stage slot must hold zeros for BOTH operands: whatever lies behind the row's end times zero is not zero when it decodes as inf / NaN)
    const int nk = KT ? (int)((kend - kbeg + 63) / 64) : (int)((kend - kbeg) / BKS);
    const int tail_chunks = KT ? (int)(((kend - kbeg) & 63) >> 3) : 0;      // 0: the last K-step is whole

    // per-lane DMA sources: piece p of this wavefront covers stage rows 8*(wid + p*NW) .. +7 (A rows first, then B rows);
    // lane l fills LDS slot (row l>>3, chunk l&7) with source chunk (l&7) ^ (l>>3)
    const bf16_t* src[PPW];
    bool beyond[PPW];      // this lane's source chunk of piece p is past the K tail
    auto point = [&](int64_t pm0, int64_t pn0) {
        const int rsub = lane / CPR;
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            const int r = (wid + p * NW) * RPP + rsub;
            if (r < BM) {
                int64_t row = pm0 + r;
                row = row < M ? row : M - 1;
                const int ch = (lane & (CPR - 1)) ^ (BKS == 64 ? (r & 7) : key32(r >> 2));
                src[p] = A + row * lda + kbeg + ch * 8;
                beyond[p] = ch >= tail_chunks;
            } else {
                const int rb = r - BM;
                int64_t row = pn0 + rb;
                row = row < N ? row : N - 1;
                // (B per row group: a tile never straddles two groups, b_group_rows % 256 == 0 is checked on the host)
                const bf16_t* Bg = epi.b_group_rows > 0 ? B + (pm0 / epi.b_group_rows) * epi.b_group_stride : B;
                const int ch = (lane & (CPR - 1)) ^ (BKS == 64 ? b_key(rb) : key32(rb >> 3));
                src[p] = Bg + row * ldb + kbeg + ch * 8;
                beyond[p] = ch >= tail_chunks;
            }
        }
    };
    point(m0, n0);
    // kstep: the K-step the stage is filled with (only the last one can be a tail)
    auto issue = [&](int stage, int kstep) {
        if (KT && tail_chunks != 0 && kstep == nk - 1) {
#pragma unroll
            for (int p = 0; p < PPW; ++p) {
                const bf16_t* s = beyond[p] ? reinterpret_cast<const bf16_t*>(dma_zero_chunk) : src[p];
                __builtin_amdgcn_global_load_lds((glb_void_ptr)s, (lds_void_ptr)(smem + stage * STAGE + (wid + p * NW) * 1024), 16, 0, 0);
                src[p] += 64;
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)src[p], (lds_void_ptr)(smem + stage * STAGE + (wid + p * NW) * 1024), 16, 0, 0);
            src[p] += BKS;
        }
    };

    f32x4 acc[FM][FN];

    // Fragment addresses.  A: row (lane & 15) of 16-row block i, chunk (4*ks + (lane >> 4)) ^ (row & 7).
    // B: fragment j of the wave tile takes the rows 32*(j >> 1) + 8*(c >> 2) + 4*(j & 1) + (c & 3), c = lane & 15, and is the FIRST
    // MFMA operand, so the accumulator is the transposed product: lane (g, c) ends up with output row 16*i + c and, over the four
    // registers of fragments (2h, 2h+1), the eight CONSECUTIVE columns 32*h + 8*g .. + 7 -- the epilogue stores 16-B vectors
    // straight from registers (four lanes = 64 contiguous bytes of a row), no LDS transpose.
    const int g = lane >> 4, c15 = lane & 15;
    // BKS = 64: ks = 0; ks = 1 flips bit 6 of the byte offset (chunk ^ 4).  BKS = 32: one k-step per stage, the key of the row's group of four
    // (rows wm * 64 + 16 i + c15: the group index is c15 >> 2 whatever i)
    const int a_sw = (BKS == 64 ? (g ^ (lane & 7)) : (g ^ key32(c15 >> 2))) * 16;
    const int a_off = (wm * (FM * 16) + c15) * RB;
    const int b_row0 = wn * (FN * 16) + 8 * (c15 >> 2) + (c15 & 3);
    // + 32*(j >> 1) + 4*(j & 1) leaves the key unchanged (BKS = 64: bits 0,1,3 of the row; BKS = 32: bits 3,4 = c15 >> 2)
    const int b_sw = (BKS == 64 ? (g ^ b_key(b_row0)) : (g ^ key32(c15 >> 2))) * 16;
    const int b_off = BM * RB + b_row0 * RB;

    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE + a_off;
        const char* sb = smem + stage * STAGE + b_off;
#pragma unroll
        for (int ks = 0; ks < BKS / 32; ++ks) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * RB + (a_sw ^ (ks * 64)));
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + ((j >> 1) * 32 + (j & 1) * 4) * RB + (b_sw ^ (ks * 64)));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    if (PERSIST) {      // (NS >= 3) prologue of the first tile; every later tile's prologue is issued in front of the previous epilogue
#pragma unroll
        for (int p = 0; p < NS - 1; ++p)
            if (p < nk) issue(p, p);
    }
    for (int vt = blockIdx.x;;) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (NS == 2) {
        issue(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int stage = kt & 1;
            if (kt + 1 < nk) {
                issue(stage ^ 1, kt + 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");      // tile kt landed; tile kt+1 stays in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();      // every wavefront's pieces of tile kt are in LDS
            asm volatile("" ::: "memory");
            compute(stage);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // every wavefront is done reading this stage before tile kt+2's DMA overwrites it
            asm volatile("" ::: "memory");
        }
    } else {
        // ring of NS stages, NS-1 tiles in flight, ONE barrier per K-step: the barrier of step kt proves both that tile kt has
        // landed for everyone and that everyone is past the fragment reads of tile kt-1, whose stage tile kt+NS-1 then takes
        if (!PERSIST) {
#pragma unroll
            for (int p = 0; p < NS - 1; ++p)
                if (p < nk) issue(p, p);
        }
        int stage = 0, fill = (NS - 1) % NS;
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = nk - 1 - kt;      // tiles issued after tile kt that may stay in flight (capped at NS-2)
            if (ahead >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
            else if (NS > 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NS - 1 < nk) issue(fill, kt + NS - 1);
            compute(stage);
            stage = stage + 1 == NS ? 0 : stage + 1;
            fill = fill + 1 == NS ? 0 : fill + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // the epilogue slab overwrites the ring
        asm volatile("" ::: "memory");
    }
    // (the barrier above: every wavefront is past its last fragment read, the whole ring is free)
    const int64_t em0 = m0, en0 = n0;
    if (PERSIST) {
        vt += gridDim.x;
        if (vt < ntiles) {
            t = xcd_remap(vt, ntiles);
            m0 = (int64_t)(t / tiles_n) * BM;
            n0 = (int64_t)(t % tiles_n) * BN;
            point(m0, n0);
#pragma unroll
            for (int p = 0; p < NS - 1; ++p)
                if (p < nk) issue(p, p);
        }
    }
    // ---- epilogue from registers: per 16-row block, two 8-column vectors per lane (N % 8 == 0 and aligned operands are
    // eligibility conditions, so there is no scalar path) ----
    const bool split = slabs != nullptr;
    float* const slab = split ? slabs + (int64_t)ksplit * M * N : nullptr;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int64_t m = em0 + wm * (FM * 16) + i * 16 + c15;
#pragma unroll
        for (int h = 0; h < FN / 2; ++h) {
            const int64_t n = en0 + wn * (FN * 16) + 32 * h + 8 * g;
            if (m < M && n < N) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = acc[i][2 * h + (u >> 2)][u & 3];
                if (split) {
                    float* dst = slab + m * N + n;
                    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(v);
                    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(v + 4);
                } else {
                    const Epi ek = epi_known<EK>(epi);
                    EpiPrefetch<TO> pf;
                    pf.load(ek, m, n, D, ldd);
                    epi_finish8<TO>(ek, v, pf, m, n, D, ldd);
                }
            }
        }
    }
    if (!PERSIST || vt >= ntiles) break;
    }      // tiles of this workgroup
}

// eligibility of a problem for the DMA pipeline (checked on the host)
inline bool dma_eligible(const iseg_gemm_args* g, int64_t kps) {
    if (!g->a_kcontig || !g->b_kcontig || g->a_act != ISEG_ACT_NONE || g->colsum_out) return false;
    // K: whole 16-B chunks per row (a last K-step of fewer than eight is zero-filled, see the kernel); a split cuts at multiples of 64
    if (g->K % 8 != 0 || (kps != g->K && kps % 64 != 0) || g->K < dma_min_k() || g->N % 8 != 0 || g->N < 64 || g->M < 64) return false;
    if (((uintptr_t)g->A % 16) || ((uintptr_t)g->B % 16) || g->lda % 8 || g->ldb % 8) return false;
    if (((uintptr_t)g->D % 16) || g->ldd % 8) return false;
    if (g->residual && (((uintptr_t)g->residual % 16) || g->ldr % 8)) return false;
    if (g->aux && (((uintptr_t)g->aux % 16) || g->ldaux % 8)) return false;
    if (g->pre_out && (((uintptr_t)g->pre_out % 16) || g->ldp % 8)) return false;
    if ((g->bias && (uintptr_t)g->bias % 16) || (g->colscale && (uintptr_t)g->colscale % 16)) return false;
    if (g->batch > 1 && (g->sa_outer % 8 || g->sa_inner % 8 || g->sb_outer % 8 || g->sb_inner % 8 || g->sd_outer % 8 || g->sd_inner % 8))
        return false;
    return true;
}

template <int WM, int WN, int NS, class TO, int FN = 4, int EK = EK_ANY, int BKS = 64, int FM = 4>
void launch_dma(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t k_per_split, float* slabs, hipStream_t s) {
    constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
    const int tiles_m = (int)ceil_div64(g->M, BM), tiles_n = (int)ceil_div64(g->N, BN);
    const int ntiles = tiles_m * tiles_n;
    const int vecD = 1;
    const int batch = g->batch > 1 ? g->batch : 1;
    dim3 grid(ntiles, nsplit, batch);
    constexpr int lds = NS * (BM + BN) * BKS * 2;
    static_assert(FM == 4 || BKS == 64, "the 32-deep stages exist for the 256 x 128 tile");
    if constexpr (BKS == 32) {      // (dispatch_dma sends only whole-K problems here: K % 32 == 0, splits cut at multiples of 64)
        static const bool raised32 = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<WM, WN, NS, TO, false, FN, EK, false, 32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        }();
        (void)raised32;
        hipLaunchKernelGGL((gemm_bf16_dma_kernel<WM, WN, NS, TO, false, FN, EK, false, 32>), grid, dim3(WM * WN * 64), lds, s, (const bf16_t*)g->A, g->lda,
                           (const bf16_t*)g->B, g->ldb, (TO*)g->D, g->ldd, g->M, g->N, g->K, tiles_n, ntiles, k_per_split, slabs, epi, vecD);
        return;
    }
    if (g->K % 64 != 0) {
        static const bool raised_kt = [] {      // > 64 KiB of dynamic LDS needs the attribute once per instantiation
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<WM, WN, NS, TO, false, FN, EK, true, 64, FM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        }();
        (void)raised_kt;
        hipLaunchKernelGGL((gemm_bf16_dma_kernel<WM, WN, NS, TO, false, FN, EK, true, 64, FM>), grid, dim3(WM * WN * 64), lds, s, (const bf16_t*)g->A, g->lda,
                           (const bf16_t*)g->B, g->ldb, (TO*)g->D, g->ldd, g->M, g->N, g->K, tiles_n, ntiles, k_per_split, slabs, epi, vecD);
        return;
    }
    static const bool raised = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<WM, WN, NS, TO, false, FN, EK, false, 64, FM>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    }();
    (void)raised;
    hipLaunchKernelGGL((gemm_bf16_dma_kernel<WM, WN, NS, TO, false, FN, EK, false, 64, FM>), grid, dim3(WM * WN * 64), lds, s, (const bf16_t*)g->A, g->lda,
                       (const bf16_t*)g->B, g->ldb, (TO*)g->D, g->ldd, g->M, g->N, g->K, tiles_n, ntiles, k_per_split, slabs, epi, vecD);
}

// persistent form of the 256 x 128 kernel: one workgroup per CU (its 144-KiB ring allows no more), each walking ntiles / gridDim.x tiles
template <class TO>
void launch_dma_persistent(const iseg_gemm_args* g, const Epi& epi, int64_t k_per_split, int cus, hipStream_t s) {
    constexpr int WM = 4, WN = 2, NS = 3, BM = WM * 64, BN = WN * 64;
    const int tiles_m = (int)ceil_div64(g->M, BM), tiles_n = (int)ceil_div64(g->N, BN);
    const int ntiles = tiles_m * tiles_n;
    constexpr int lds = NS * (BM + BN) * 128;
    static const bool raised = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<WM, WN, NS, TO, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    }();
    (void)raised;
    const int grid = (ntiles < cus ? ntiles : cus) & ~7;      // a multiple of 8: every virtual id of a workgroup sits on its XCD
    hipLaunchKernelGGL((gemm_bf16_dma_kernel<WM, WN, NS, TO, true>), dim3(grid), dim3(WM * WN * 64), lds, s, (const bf16_t*)g->A, g->lda,
                       (const bf16_t*)g->B, g->ldb, (TO*)g->D, g->ldd, g->M, g->N, g->K, tiles_n, ntiles, k_per_split, (float*)nullptr, epi, 1);
}

int dma_mode();      // ISEG_GEMM_DMA: 0 = never, 1 = whenever eligible (default)

// 1: 128 x 64 (4-deep ring)   2: 256 x 128 (8 wavefronts, 3-deep ring)   3: 128 x 128, two workgroups per CU (2 stages)
// 4: 128 x 128, one workgroup per CU with the 3-deep ring.  (dispatch_dma adds two forms of 2: persistent, and 256 x 192.)  Measured on MI355X (tools/kbench_gemm_ref.py): 256 x 128 once it fills
// most CUs -- it reads each B panel half as often; otherwise 128 x 128, two per CU when there are enough tiles, the deeper ring when
// the grid is thin.
inline int dma_variant(const iseg_gemm_args* g, int nsplit) {
    // experiment knobs.  Measured on the flagship step: 128 x 128 tiles, two workgroups per CU (3) for the GEMMs with fused gelu / aux / residual
    // epilogues 10.66 vs 10.69 ms (equal), the deep-ring 128 x 128 (4) 11.28 ms; requesting the fused operands of four row blocks together
    // in the epilogue: equal, +44 registers.  The 13-17 us these epilogues add are VALU (gelu) and un-overlapped operand reads either way.
    static const int forced = [] { const char* e = getenv("ISEG_GEMM_DMA_VARIANT"); return e ? atoi(e) : 0; }();
    static const int forced_epi = [] { const char* e = getenv("ISEG_GEMM_DMA_VARIANT_EPI"); return e ? atoi(e) : 0; }();
    if (forced >= 1 && forced <= 4 && g->N > 64) return forced;
    if (forced_epi >= 1 && forced_epi <= 4 && g->N > 64 && (g->act != 0 || g->aux || g->residual)) return forced_epi;
    const int64_t tiles256 = ceil_div64(g->M, 256) * ceil_div64(g->N, 128), tiles128 = ceil_div64(g->M, 128) * ceil_div64(g->N, 128);
    if (g->N <= 64) return 1;
    static const int t256 = [] { const char* e = getenv("ISEG_GEMM_DMA_T256"); return e ? atoi(e) : 192; }();
    if (tiles256 * nsplit >= t256) return 2;
    if (tiles128 * nsplit >= 384) return 3;
    return 4;
}

// number of CUs of the current device (persistent grids)
inline int dma_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
        return v;
    }();
    return n;
}

// dma_variant plus the two further forms of the 256-row tile:
//   6: 256 x 192 tiles (a wavefront owns 64 x 96, two ring stages = 112 KiB): a third fewer re-reads of the A panel than 256 x 128 when N is a
//      multiple of 192 (the 4C = 768 / 1536 / 3072 outputs of the ConvNeXt stages); taken when its tiles fill whole rounds of the CUs better
//      (with equal rounds the flagship step measured the same and the in-situ launches of the x aux data gradient 51.7 vs 50.5 us).
//      Measured (tools/kbench_gemm_dma_ab.py): M=4096 N=3072 K=768 34.5 -> 27.5 us (384 tiles = 1.5 rounds -> 256 = one round),
//      M=16384 N=1536 K=384 34.4 -> 33.1 us, with the x aux epilogue 42.2 -> 41.4 us, M=65536 N=768 K=192 45.7 -> 43.0 us.
//   5: several 256 x 128 tiles per CU, one K split, no batch: the persistent form overlaps a tile's epilogue with the next tile's first DMAs
//      (ISEG_GEMM_DMA_PERSIST=1; see below why it is not the default).
inline int dma_form(const iseg_gemm_args* g, int nsplit) {
    const int variant = dma_variant(g, nsplit);
    // 9 (round 6): 64 x 192 tiles (eight wavefronts of 16 x 96) where the thin-grid 128 x 128 form leaves a quarter of the CUs idle --
    // M = 4096, N = 768 (the K = 3072 products of stage 3): 256 workgroups instead of 192; + bias + residual 40.0 -> 34.7 us, plain 36.0 -> 32.4 us
    static const int flat = [] { const char* e = getenv("ISEG_GEMM_DMA_64X192"); return e ? atoi(e) : 1; }();
    if (flat && variant == 4 && nsplit == 1 && g->batch <= 1 && g->N % 192 == 0 && g->K % 64 == 0) {
        const int64_t t4 = ceil_div64(g->M, 128) * ceil_div64(g->N, 128), t9 = ceil_div64(g->M, 64) * (g->N / 192);
        if (t4 * 8 <= (int64_t)dma_cus() * 7 && t9 <= dma_cus() && t9 * 6 >= t4 * 7) return 9;
    }
    if (variant != 2 || nsplit != 1 || g->batch > 1) return variant;
    static const int wide = [] { const char* e = getenv("ISEG_GEMM_DMA_WIDE"); return e ? atoi(e) : 1; }();
    // off by default: 1 us per launch faster back to back (tools/kbench_gemm_ref.py) but slower inside the training step -- 59.8 vs 53.9 us for the
    // x aux data gradient under bench.py's event timer, 9.82 vs 9.70 ms per step: three tiles per workgroup is a static partition, and the
    // hardware dispatcher's dynamic one (768 workgroups, a new one wherever a CU frees up) absorbs CUs that run behind on cold operands
    static const int persist = [] { const char* e = getenv("ISEG_GEMM_DMA_PERSIST"); return e ? atoi(e) : 0; }();
    const int cus = dma_cus();
    const int64_t t128 = ceil_div64(g->M, 256) * ceil_div64(g->N, 128);
    // 8 (round 6): 128 x 192 tiles where the 256 x 128 ones leave more than an eighth of the CUs idle in their single round and these fit one
    // round with at least a sixth more workgroups -- M = 16384, N = 384 (the K = 1536 products of stage 2): 256 workgroups instead of 192
    static const int tall = [] { const char* e = getenv("ISEG_GEMM_DMA_128X192"); return e ? atoi(e) : 1; }();
    if (tall && g->N % 192 == 0 && t128 * 8 <= (int64_t)cus * 7) {
        const int64_t t8 = ceil_div64(g->M, 128) * (g->N / 192);
        if (t8 <= cus && t8 * 6 >= t128 * 7) return 8;
    }
    if (wide && g->N % 192 == 0) {
        const int64_t t192 = ceil_div64(g->M, 256) * (g->N / 192);
        const double e128 = (double)t128 / (double)(ceil_div64(t128, cus) * cus), e192 = (double)t192 / (double)(ceil_div64(t192, cus) * cus);
        if (t192 >= cus && e192 >= e128 + 0.1) return 6;      // only where it fills the rounds better: equal rounds measured equal in situ
    }
    if (persist && t128 > cus && g->K % 64 == 0) return 5;      // (the persistent instantiation has no K-tail form)
    return variant;
}

// one instantiation per fused epilogue kind for bf16 outputs (the other output type keeps the run-time epilogue)
template <int WM, int WN, int NS, class TO, int FN, int BKS = 64, int FM = 4>
void launch_dma_kinds(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    if (sizeof(TO) == 2) {
        switch (epi_kind(epi, slabs)) {
            case EK_GELU_DERIV: launch_dma<WM, WN, NS, TO, FN, EK_GELU_DERIV, BKS, FM>(g, epi, nsplit, kps, slabs, s); return;
            case EK_MUL_AUX: launch_dma<WM, WN, NS, TO, FN, EK_MUL_AUX, BKS, FM>(g, epi, nsplit, kps, slabs, s); return;
            case EK_BIAS_RESIDUAL: launch_dma<WM, WN, NS, TO, FN, EK_BIAS_RESIDUAL, BKS, FM>(g, epi, nsplit, kps, slabs, s); return;
            case EK_PLAIN: launch_dma<WM, WN, NS, TO, FN, EK_PLAIN, BKS, FM>(g, epi, nsplit, kps, slabs, s); return;
            case EK_BIAS: launch_dma<WM, WN, NS, TO, FN, EK_BIAS, BKS, FM>(g, epi, nsplit, kps, slabs, s); return;
            default: break;
        }
    }
    launch_dma<WM, WN, NS, TO, FN, EK_ANY, BKS, FM>(g, epi, nsplit, kps, slabs, s);
}

// ISEG_GEMM_DMA_BK32: the 256 x 128 form with 32-deep ring stages, two workgroups per CU (see the kernel's BKS note): 1 (default) = for K <= 512,
// 2 = every whole-K problem of the 256 x 128 form, 0 = never
int dma_bk32();

template <class TO>
void dispatch_dma(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    switch (dma_form(g, nsplit)) {
        case 1: launch_dma<2, 1, 4, TO>(g, epi, nsplit, kps, slabs, s); break;
        case 2:      // 256 x 128: the flagship's stage-2 products
            // measured (tools/kbench_pitch.py, M = 16384, one box): K = 384 with the gelu + gelu' epilogue 44.8 -> 41.2 us, with the x aux epilogue
            // 39.7 -> 35.7; K = 1536 plain 31.8 -> 34.0, + bias + residual 35.6 -> 34.9 -- the short-K products are epilogue-bound and gain from the
            // second resident workgroup, the long-K ones pay for twice the barriers: taken for K <= 512 (ISEG_GEMM_DMA_BK32 = 0 never, 2 always)
            // (128 x 128 tiles with 48 KB rings, THREE workgroups per CU, measured 45.4 / 38.6 us on the two K = 384 launches: the extra fill
            // traffic of the smaller tile costs more than the third resident workgroup returns)
            if (dma_bk32() && (dma_bk32() == 2 || g->K <= 512) && g->K % 32 == 0 && g->K >= 96 && (kps == g->K || kps % 64 == 0) && sizeof(TO) == 2)
                launch_dma_kinds<4, 2, 3, TO, 4, 32>(g, epi, nsplit, kps, slabs, s);
            else launch_dma_kinds<4, 2, 3, TO, 4>(g, epi, nsplit, kps, slabs, s);
            break;
        case 3: launch_dma<2, 2, 2, TO>(g, epi, nsplit, kps, slabs, s); break;
        case 5: launch_dma_persistent<TO>(g, epi, kps, dma_cus(), s); break;
        case 6: launch_dma_kinds<4, 2, 2, TO, 6>(g, epi, nsplit, kps, slabs, s); break;      // 256 x 192: stage 3
        case 9: launch_dma_kinds<4, 2, 3, TO, 6, 64, 1>(g, epi, nsplit, kps, slabs, s); break;      // 64 x 192, eight wavefronts of 16 x 96
        case 8: launch_dma_kinds<4, 2, 3, TO, 6, 64, 2>(g, epi, nsplit, kps, slabs, s); break;      // 128 x 192, eight wavefronts of 32 x 96
        default: launch_dma_kinds<2, 2, 3, TO, 4>(g, epi, nsplit, kps, slabs, s); break;
    }
}

}  // namespace iseg_mm
