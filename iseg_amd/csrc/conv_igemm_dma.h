// Implicit-GEMM convolution on the LDS-DMA pipeline (forward and data gradient, one group, channel counts that are multiples of 64).
//
// Same ring / fragment / register-epilogue scheme as gemm_bf16_dma_kernel (gemm_dma.h: global_load_lds_dwordx4 into NS unpadded, XOR-swizzled
// stages, counted vmcnt waits, raw barriers, B as the first MFMA operand so a lane ends with eight consecutive output columns) -- what differs
// is where a DMA piece's per-lane SOURCE address comes from.  A K-step is 64 channels of ONE tap (Cg % 64 == 0), so per K-step and stage row:
//   pass 0 (forward)    A row = output pixel (n, oh, ow): x[n, oh*sh + i*dh - pt, ow*sw + j*dw - pl, c0 .. c0+63], or a zero page in TF's halo
//                       B row = output channel of the K-contiguous kernel copy Wt [Cout][kh*kw*Cin] (nn.wt): a plain K-advancing pointer
//   pass 1 (data grad)  A row = input pixel (n, h, w): dy[n, (h + pt - i*dh)/sh, (w + pl - j*dw)/sw, co0 .. co0+63] where both quotients are
//                       exact and inside the map, else the zero page;  B row = input channel c of tap (i, j): w[i, j, c, co0 .. co0+63]
// The register-staged gather of conv_igemm.hip spends its issue slots on address arithmetic, two LDS writes per chunk and the staging
// registers; here the address of a piece is ~10 VALU per K-step and the data never touches a register.  The ASPP 3x3 convolutions of the
// flagship (16 x 16 x 16 x 768 -> 256, K = 6912, 8 K-splits) ran 47.5 / 67.3 us (forward / data gradient): see DESIGN 5 for what this form gives.
#pragma once
#include "gemm_dma.h"

namespace {

__device__ uint4 conv_zero_page[8];      // 128 zero bytes (device globals are zero-initialised): one swizzled stage row of a halo tap

template <int WM, int WN, int NS, int PASS, class TO>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_dma_kernel(ConvP p, const bf16_t* __restrict__ Bop, int64_t ldb, int64_t tap_stride,
                                                                      TO* __restrict__ D, int64_t ldd, int64_t M, int64_t N, int64_t K,
                                                                      int tiles_n, int ntiles, int64_t k_per_split, float* __restrict__ slabs,
                                                                      Epi epi) {
    using namespace iseg_mm;
    constexpr int FN = 4;
    constexpr int NW = WM * WN;
    constexpr int BM = WM * 64, BN = WN * FN * 16;
    constexpr int PIECES = (BM + BN) / 8, PPW = PIECES / NW;
    static_assert(PIECES % NW == 0, "stage pieces must divide over the wavefronts");
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    int t, ksplit;
    tile_and_split(ntiles, t, ksplit);
    const int64_t m0 = (int64_t)(t / tiles_n) * BM, n0 = (int64_t)(t % tiles_n) * BN;
    const int64_t kbeg = (int64_t)ksplit * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    const int nk = (int)((kend - kbeg) / 64);

    // per piece: the row-side decode (fixed over the reduction).  A rows: (a0, a1) = tap origin / pixel + padding, a2 = image base (pixels);
    // B rows: element offset of the row inside one tap (pass 1) or of the row in Wt (pass 0)
    int a0[PPW], a1[PPW];
    int64_t a2[PPW];
    const int rsub = lane >> 3;
    const bf16_t* const zero = reinterpret_cast<const bf16_t*>(conv_zero_page);
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int r = (wid + q * NW) * 8 + rsub;
        if (r < BM) {
            int64_t row = m0 + r;
            row = row < M ? row : M - 1;
            const int hw = p.Hr * p.Wr, n = (int)(row / hw), rem = (int)(row % hw);
            if (PASS == 0) {
                a0[q] = (rem / p.Wr) * p.sh - p.pt;
                a1[q] = (rem % p.Wr) * p.sw - p.pl;
            } else {
                a0[q] = rem / p.Wr + p.pt;
                a1[q] = rem % p.Wr + p.pl;
            }
            a2[q] = (int64_t)n * p.Hs * p.Ws;
        } else {
            const int rb = r - BM;
            int64_t row = n0 + rb;
            row = row < N ? row : N - 1;
            a0[q] = ((lane & 7) ^ b_key(rb)) * 8;      // swizzled chunk of this lane
            a1[q] = 0;
            a2[q] = row * ldb;
        }
    }
    auto issue = [&](int stage, int kt) {
        const int64_t k0 = kbeg + (int64_t)kt * 64;
        const int tap = (int)(k0 / p.Cg), c0 = (int)(k0 % p.Cg);      // (uniform)
        const int ti = tap / p.kw, tj = tap % p.kw;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int r = (wid + q * NW) * 8 + rsub;
            const bf16_t* src;
            if (r < BM) {
                const int chunk = ((lane & 7) ^ (r & 7)) * 8;
                int ih, iw;
                bool ok;
                if (PASS == 0) {
                    ih = a0[q] + ti * p.dh;
                    iw = a1[q] + tj * p.dw;
                    ok = (unsigned)ih < (unsigned)p.Hs && (unsigned)iw < (unsigned)p.Ws;
                } else {
                    const int th = a0[q] - ti * p.dh, tw = a1[q] - tj * p.dw;
                    ih = th / p.sh;
                    iw = tw / p.sw;
                    ok = th >= 0 && tw >= 0 && ih * p.sh == th && iw * p.sw == tw && ih < p.Hs && iw < p.Ws;
                }
                src = ok ? p.src + (a2[q] + (int64_t)ih * p.Ws + iw) * p.Cs + c0 + chunk : zero + chunk;
            } else if (PASS == 0) {
                src = Bop + a2[q] + k0 + a0[q];
            } else {
                src = Bop + (int64_t)tap * tap_stride + a2[q] + c0 + a0[q];
            }
            __builtin_amdgcn_global_load_lds((glb_void_ptr)src, (lds_void_ptr)(smem + stage * STAGE + (wid + q * NW) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][FN];
    const int g = lane >> 4, c15 = lane & 15;
    const int a_sw = (g ^ (lane & 7)) * 16;
    const int a_off = (wm * 64 + c15) * 128;
    const int b_row0 = wn * (FN * 16) + 8 * (c15 >> 2) + (c15 & 3);
    const int b_sw = (g ^ b_key(b_row0)) * 16;
    const int b_off = BM * 128 + b_row0 * 128;
    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE + a_off;
        const char* sb = smem + stage * STAGE + b_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[FN];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 2048 + (a_sw ^ (ks * 64)));
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + ((j >> 1) * 32 + (j & 1) * 4) * 128 + (b_sw ^ (ks * 64)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (NS == 2) {
        issue(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int stage = kt & 1;
            if (kt + 1 < nk) {
                issue(stage ^ 1, kt + 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            compute(stage);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    } else {
#pragma unroll
        for (int q = 0; q < NS - 1; ++q)
            if (q < nk) issue(q, q);
        int stage = 0, fill = (NS - 1) % NS;
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = nk - 1 - kt;
            if (ahead >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NS - 1 < nk) issue(fill, kt + NS - 1);
            compute(stage);
            stage = stage + 1 == NS ? 0 : stage + 1;
            fill = fill + 1 == NS ? 0 : fill + 1;
        }
    }
    // ---- epilogue from registers (as gemm_bf16_dma_kernel) ----
    const bool split = slabs != nullptr;
    float* const slab = split ? slabs + (int64_t)ksplit * M * N : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + c15;
#pragma unroll
        for (int h = 0; h < FN / 2; ++h) {
            const int64_t n = n0 + wn * (FN * 16) + 32 * h + 8 * g;
            if (m < M && n < N) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = acc[i][2 * h + (u >> 2)][u & 3];
                if (split) {
                    float* dst = slab + m * N + n;
                    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(v);
                    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(v + 4);
                } else {
                    EpiPrefetch<TO> pf;
                    pf.load(epi, m, n, D, ldd);
                    epi_finish8<TO>(epi, v, pf, m, n, D, ldd);
                }
            }
        }
    }
}

}  // namespace
