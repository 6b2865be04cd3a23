// Weight gradients of the fused ConvNeXt MLP (backbones/convnext.py:51-63 of the reference) without an [M, 4C] tensor in HBM.
//
// The backward chain kernel of mlp_fused.hip used to write G = gelu(H) and dH (two [M, 4C] bf16 tensors, 400 MB at stage 0) only so that
// two weight-gradient GEMMs could read them back.  Here the hidden tile is recomputed a second time, by workgroups that own a SLICE OF THE
// HIDDEN UNITS and walk over a chunk of rows:
//     H[m][hid]  = y2[m][:] . W1[:][hid] + b1[hid]          (16x16x32 MFMA: A = y2 rows from the LDS tile, B = W1 fragments in registers)
//     dG[m][hid] = dbr[m][:] . (W2 gamma)[hid][:]           (A = dbr rows, B = W2 gamma fragments in registers)
//     G = gelu(H),  dH = dG o gelu'(H)                      (registers; rounded to bf16 as in the chain kernel)
//     Z[hid][c]    += sum_m G[m][hid]  dbr[m][c]            (= g^T dbr:  dW2 = Z gamma, dgamma = sum_hid W2 o Z + b2 S)
//     dW1T[hid][c] += sum_m dH[m][hid] y2[m][c]             (= (y2^T dH)^T)
//     db1[hid]     += sum_m dH[m][hid],     S[c] += sum_m dbr[m][c]
// A wavefront owns 16 hidden units.  Its H / dG accumulators (lane = hidden unit, registers = rows) ARE the A operand of the two
// row-contractions: accumulator blocks of rows 0-15 and 16-31 side by side are one 8-element fragment whose k slot (q, j) is row 4q + j
// (j < 4) or 16 + 4q + j - 4, and the B operand (lane = channel) is gathered in that same order from the row-major LDS tile by two
// ds_read_b64_tr_b16.  Z and dW1T stay in registers (C / 2 per lane) over the whole row chunk; every workgroup writes one fp32 partial,
// a fixed-order row reduction sums the chunk records (no atomics) and a small finish launch applies the layer-scale algebra.
//
// Measured (MI355X, stage 0: M = 262144, C = 96): 165 us + 30 us for the finish launch; the instruction stream is VALU-issue bound (141 VALU + 24
// MFMA per 32-row block and wavefront, gelu + gelu' = 110 of them; SQ issue ~85 % busy with two wavefronts per SIMD), LDS 31 us, matrix cores
// 31 us.  What did not help: requesting the next block's A fragments a phase early (202 vs 198 us), 128-row tiles (222 us).
//
// dbr = rowscale[sample] * d(out) is formed while the tile is staged (the drop-path row factor is constant inside a tile), and with
// mean != NULL the y operand is LayerNorm(y1) formed the same way, so neither the scaled gradient nor the normalised activation has to
// exist in HBM for this kernel.
#include "common.h"
#include "iseg_hip.h"
#include "mlp_common.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

template <int C> struct WgGeom {
    static constexpr int WAVES = 8, NT = 64 * WAVES, RT = C > 96 ? 32 : 64;      // rows per LDS tile (registers: C / 2 accumulators + C / 4 weights per lane)
    static constexpr int HID = 4 * C, KS = C / 32, CBS = C / 16;
    static constexpr int NHG = HID / (16 * WAVES);                       // workgroups (hidden groups) per row chunk
    // tile row stride: 32 bytes past a multiple of 256.  Conflict-free for both kinds of read: the ds_read_b128 lane groups see slots
    // (2 row + k-quarter) mod 16, even for one half of a group and odd for the other; a 32-lane half of the ds_read_b64_tr_b16 touches 8 rows x
    // 32 bytes at 8-dword row steps = the 64 banks once (SQ_LDS_BANK_CONFLICT was 48 % of the LDS cycles with a 16-byte pad)
    static constexpr int STRIDE = (((2 * C + 223) / 256) * 256 + 32) / 2;
    static constexpr int TILE = RT * STRIDE;                             // elements per tile and tensor
    static constexpr int CPR = C / 8;                                    // 16-byte chunks per row
    static constexpr int RG = (NT / CPR) >= 32 ? 32 : 16;                // staging row groups
    static constexpr int NLOAD = CPR * RG, RPT = RT / RG;                // staging threads, rows per thread and tile
    static constexpr int LDS = 2 * 2 * TILE * 2;                         // two buffers x (y, d)
    static constexpr int IMG = C * 64, SLAB = 3 * IMG;                   // tiled backward weight buffer of mlp_fused.hip: [slab][A1 | A3 | A4]
    static_assert(C % 32 == 0 && HID % (16 * WAVES) == 0 && NLOAD <= NT && RT % RG == 0 && LDS <= 160 * 1024, "geometry");
};

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// B fragment (lane = channel 16 cb + (lane & 15)) of the 32 tile rows at `rows` in the accumulator's row order: k slot (q, j) = row 4 q + j
// (j < 4), 16 + 4 q + j - 4 (j >= 4)
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* rows, int stride, int cb, int lane) {
    const int q = lane >> 4, qq = (lane >> 2) & 3, p = lane & 3;
    const bf16_t* a0 = rows + (4 * q + qq) * stride + 16 * cb + 4 * p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 16 * stride));
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = lo[i];
        f[4 + i] = hi[i];
    }
    return f;
}

template <int C>
__global__ __launch_bounds__(512) void convnext_mlp_wgrad_kernel(const bf16_t* __restrict__ Y, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ ln_gamma,
                                                                 const float* __restrict__ ln_beta, const bf16_t* __restrict__ D,
                                                                 const float* __restrict__ rowscale, int64_t rows_per_group,
                                                                 const void* __restrict__ BW, const float* __restrict__ b1,
                                                                 float* __restrict__ part, int64_t M, int64_t rows_per_chunk, int nchunk) {
    using G = WgGeom<C>;
    constexpr int HID = G::HID, KS = G::KS, CBS = G::CBS, STRIDE = G::STRIDE, TILE = G::TILE, RT = G::RT, RG = G::RG, RPT = G::RPT, CPR = G::CPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* const tiles = reinterpret_cast<bf16_t*>(smem);      // [buf][y | d][RT][STRIDE]

    // XCD-aware placement: consecutive workgroup ids go round the 8 XCDs, so the NHG hidden groups of one row chunk (which read the same
    // rows) are given ids that land on ONE XCD and share its L2
    const int wg = blockIdx.x, xcd = wg & 7, idx = wg >> 3;
    const int hg = idx % G::NHG, chunk = (idx / G::NHG) * 8 + xcd;
    if (chunk >= nchunk) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int hid0 = (hg * G::WAVES + wid) * 16;                 // this wavefront's 16 hidden units
    const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
    const int64_t r_end = r_begin + rows_per_chunk < M ? r_begin + rows_per_chunk : M;
    const int ntiles = (int)((r_end - r_begin + RT - 1) / RT);

    // ---- weight fragments of the 16 hidden units, all C input channels: B operands of the two recomputed products, kept in registers ----
    bf16x8 w1f[KS], w3f[KS];
    {
        const char* slab = (const char*)BW + (int64_t)(hid0 >> 5) * G::SLAB;
        const int row = (hid0 & 16) + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int off = (2 * ks + (q >> 1)) * 1024 + mlp_frag_offset(row, q & 1);
            w1f[ks] = *reinterpret_cast<const bf16x8*>(slab + off);
            w3f[ks] = *reinterpret_cast<const bf16x8*>(slab + G::IMG + off);
        }
    }
    const float bias = b1[hid0 + li];

    // ---- staging role: thread -> (row group, 16-byte chunk of the row); the chunk position is fixed so S can ride in registers ----
    const bool loader = tid < G::NLOAD;
    const int cpos = tid % CPR, rg = loader ? tid / CPR : 0;      // (the other threads load row group 0 again and drop it)
    float s_acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s_acc[u] = 0.f;
    bf16x8 ry[RPT], rd[RPT];
    float rmean[RPT], rrstd[RPT];
    bool rok[RPT];

    // Global loads of tile t into registers.  No branch: past the last tile the loads repeat the last tile's addresses and every row is
    // flagged invalid (zeros go to the spare LDS buffer) -- with `if (t + 1 < ntiles)` around issue() and commit() hipcc sinks the loads
    // down into the commit block, i.e. behind the compute phase they were meant to overlap.
    auto issue = [&](int t) {
        const bool live = t < ntiles;
        const int tt = live ? t : ntiles - 1;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int64_t row = r_begin + (int64_t)tt * RT + rg + RG * i;
            const bool ok = live && loader && row < r_end;
            const int64_t rr = row < r_end ? row : r_begin;
            ry[i] = *reinterpret_cast<const bf16x8*>(Y + rr * C + 8 * cpos);
            rd[i] = *reinterpret_cast<const bf16x8*>(D + rr * C + 8 * cpos);
            if (mean) {
                rmean[i] = mean[rr];
                rrstd[i] = rstd[rr];
            }
            rok[i] = ok;      // (applied in commit(): touching the loaded registers here would put the wait in front of the compute phase)
        }
    };
    auto commit = [&](int t, int buf) {      // registers -> LDS tile `buf`: row factor, LayerNorm, column sums of dbr
        const int64_t row0 = r_begin + (int64_t)(t < ntiles ? t : ntiles - 1) * RT;
        const float rs = rowscale ? rowscale[row0 / rows_per_group] : 1.f;
        bf16_t* const ty = tiles + (buf * 2 + 0) * TILE;
        bf16_t* const td = tiles + (buf * 2 + 1) * TILE;
        float lg[8], lb[8];      // (re-read per tile from L1: 16 registers less over the main loop)
        if (mean) {
            load8<float>(ln_gamma + 8 * cpos, lg);
            load8<float>(ln_beta + 8 * cpos, lb);
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = rg + RG * i;
            bf16x8 y = ry[i], d = rd[i];
            if (!rok[i]) {      // rows past the chunk: zero dbr (no contribution anywhere), finite y
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    y[u] = (bf16_t)0.f;
                    d[u] = (bf16_t)0.f;
                }
            }
            if (mean && rok[i]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) y[u] = (bf16_t)iseg_ln_apply((float)y[u], rmean[i], rrstd[i], lg[u], lb[u]);
            }
            if (rowscale) {
#pragma unroll
                for (int u = 0; u < 8; ++u) d[u] = (bf16_t)((float)d[u] * rs);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s_acc[u] += (float)d[u];
            if (loader) {
                *reinterpret_cast<bf16x8*>(ty + r * STRIDE + 8 * cpos) = y;
                *reinterpret_cast<bf16x8*>(td + r * STRIDE + 8 * cpos) = d;
            }
        }
    };

    f32x4 zacc[CBS], wacc[CBS];
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) zacc[cb][r] = wacc[cb][r] = 0.f;
    float db1 = 0.f;

    // Per 32-row block: every A fragment of the recomputed products is requested before their MFMAs, and the transposed B fragments of
    // the row contractions are requested BEFORE the gelu section, whose VALU time covers their latency (left alone hipcc sinks each
    // ds_read next to its MFMA and waits for it: 36 exposed LDS round trips per block, 7x the time).
    constexpr int KG = KS > 3 ? 2 : KS, CG = CBS > 6 ? 4 : CBS;      // fragment groups that fit the register budget at C = 192
    auto compute = [&](int buf) {
        const bf16_t* const ty = tiles + (buf * 2 + 0) * TILE;
        const bf16_t* const td = tiles + (buf * 2 + 1) * TILE;
#pragma unroll
        for (int mb = 0; mb < RT / 32; ++mb) {
            f32x4 hacc[2], dacc[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hacc[b][r] = bias;
                    dacc[b][r] = 0.f;
                }
#pragma unroll
            for (int k0 = 0; k0 < KS; k0 += KG) {
                bf16x8 ay[KG][2], ad[KG][2];
#pragma unroll
                for (int k = 0; k < KG; ++k)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int o = (mb * 32 + b * 16 + li) * STRIDE + 32 * (k0 + k) + 8 * q;
                        ay[k][b] = *reinterpret_cast<const bf16x8*>(ty + o);
                        ad[k][b] = *reinterpret_cast<const bf16x8*>(td + o);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < KG; ++k)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        hacc[b] = mfma16(ay[k][b], w1f[k0 + k], hacc[b]);
                        dacc[b] = mfma16(ad[k][b], w3f[k0 + k], dacc[b]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            bf16x8 bd[CG], by[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                bd[c] = tr_frag(td + mb * 32 * STRIDE, STRIDE, c, lane);
                by[c] = tr_frag(ty + mb * 32 * STRIDE, STRIDE, c, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 gf, hf;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float g, gd;
                    gelu_sig_both(hacc[b][r], g, gd);
                    const bf16_t dh = (bf16_t)(dacc[b][r] * gd);
                    gf[4 * b + r] = (bf16_t)g;
                    hf[4 * b + r] = dh;
                    db1 += (float)dh;
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c0 = 0; c0 < CBS; c0 += CG) {
                if (c0 > 0) {
#pragma unroll
                    for (int c = 0; c < CG; ++c) {
                        bd[c] = tr_frag(td + mb * 32 * STRIDE, STRIDE, c0 + c, lane);
                        by[c] = tr_frag(ty + mb * 32 * STRIDE, STRIDE, c0 + c, lane);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    zacc[c0 + c] = mfma16(gf, bd[c], zacc[c0 + c]);
                    wacc[c0 + c] = mfma16(hf, by[c], wacc[c0 + c]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    issue(0);
    commit(0, 0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        issue(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(t & 1);
        commit(t + 1, (t + 1) & 1);
        __syncthreads();
    }

    // ---- partial results of this (chunk, hidden group) ----
    // one record per chunk: [Z | dW1T | db1 | S], so that ONE fixed-order row reduction sums everything over the chunks
    constexpr int64_t REC = 2 * (int64_t)HID * C + HID + C;
    float* const pz = part + (int64_t)chunk * REC;
    float* const pw = pz + (int64_t)HID * C;
    float* const pb = pw + (int64_t)HID * C;
    float* const ps = pb + HID;
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t o = (int64_t)(hid0 + 4 * q + r) * C + 16 * cb + li;
            pz[o] = zacc[cb][r];
            pw[o] = wacc[cb][r];
        }
    {
        // the four lanes li, li + 16, li + 32, li + 48 hold the same hidden unit
        unsigned u = __builtin_bit_cast(unsigned, db1);
        auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        float v = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
        u = __builtin_bit_cast(unsigned, v);
        auto sx = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)sx[0]) + __builtin_bit_cast(float, (unsigned)sx[1]);
        if (lane < 16) pb[hid0 + li] = v;
    }
    if (hg == 0) {      // S = column sums of dbr: the staging threads' registers -> LDS [row group][C] -> one value per channel
        float* const ss = reinterpret_cast<float*>(smem);      // (every wavefront is past its last tile read: the loop ended on a barrier)
        if (loader) {
#pragma unroll
            for (int u = 0; u < 8; ++u) ss[rg * C + 8 * cpos + u] = s_acc[u];
        }
        __syncthreads();
        if (tid < C) {
            float v = 0.f;
            for (int g = 0; g < RG; ++g) v += ss[g * C + tid];
            ps[tid] = v;
        }
    }
}

// Book the parameter gradients from the chunk-summed record (accumulating into the flat gradient buffer):
//     dW1[c][hid] += dW1T[hid][c]      db1[hid] += ..      dW2[hid][c] += Z[hid][c] gamma[c]      db2[c] += gamma[c] S[c]
//     dgamma[c]   += sum_hid W2[hid][c] Z[hid][c] + b2[c] S[c]           (gamma == NULL: dW2 += Z, db2 += S)
// Block = 8 hidden units x 32 channels (256 threads, one element each); dW1 leaves through an LDS transpose; the per-block column sums of
// W2 o Z go to `gpart` [HID / 8][C] for the fixed-order second stage.  (Summing the 80 chunk records inside this kernel took 47-134 us in
// three different shapes -- 144 workgroups cannot pull 24 MB, and the db1 / S sums were 80 dependent loads on a handful of threads; the
// generic row reduction in front does it in 8.5 us.)
constexpr int FIN_ROWS = 8;
// `rec` = the chunk-summed record [Z | dW1T | db1 | S] (fp32, one fixed-order row reduction over the chunk records in front of this launch)
__global__ __launch_bounds__(256) void convnext_mlp_wgrad_finish_kernel(const float* __restrict__ rec, const float* __restrict__ W2,
                                                                        const float* __restrict__ b2, const float* __restrict__ gamma,
                                                                        float* __restrict__ dW1, float* __restrict__ db1,
                                                                        float* __restrict__ dW2, float* __restrict__ db2,
                                                                        float* __restrict__ gpart, int C) {
    __shared__ float tw[FIN_ROWS][33];
    __shared__ float tg[8][32];
    const int HID = 4 * C;
    const int c0 = blockIdx.x * 32, h0 = blockIdx.y * FIN_ROWS;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int c = c0 + tx;
    const int64_t plane = (int64_t)HID * C;
    const float gm = gamma ? gamma[c] : 1.f;
    {
        const int64_t o = (int64_t)(h0 + ty) * C + c;
        const float z = rec[o], w = rec[plane + o];
        dW2[o] += z * gm;
        tg[ty][tx] = W2[o] * z;
        tw[ty][tx] = w;
    }
    __syncthreads();
    {      // dW1[c][hid]: thread -> (channel row, hidden unit): FIN_ROWS consecutive hidden units per channel
        const int hx = threadIdx.x & (FIN_ROWS - 1), cc = threadIdx.x / FIN_ROWS;      // 8 x 32
        dW1[(int64_t)(c0 + cc) * HID + h0 + hx] += tw[hx][cc];
    }
    if (ty == 0) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += tg[j][tx];
        if (blockIdx.y == 0) {
            const float s = rec[2 * plane + HID + c];
            if (gamma) {
                v += b2[c] * s;
                db2[c] += gm * s;
            } else {
                db2[c] += s;
            }
        }
        if (gamma) gpart[(int64_t)blockIdx.y * C + c] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < FIN_ROWS) {
        const int hid = h0 + threadIdx.x;
        db1[hid] += rec[2 * plane + hid];
    }
}

template <int C> int rows_per_chunk_for(int64_t M, int* nchunk) {
    using G = WgGeom<C>;
    // One workgroup per CU in ONE resident round: the kernel's ids go round the 8 XCDs (32 CUs each), so an XCD gets NHG * ceil(nchunk / 8)
    // workgroups -- at most 32, or the 33rd waits for a whole workgroup lifetime (84 chunks at C = 96 measured 287 us instead of ~125).
    const int target = 8 * (32 / G::NHG);
    int64_t rpc = (M + target - 1) / target;
    rpc = (rpc + 63) / 64 * 64;
    if (rpc < 256) rpc = 256;
    *nchunk = (int)((M + rpc - 1) / rpc);
    return (int)rpc;
}

template <int C>
int launch_wgrad(const void* y, const float* mean, const float* rstd, const float* lng, const float* lnb, const void* d, const float* rowscale,
                 int64_t rows_per_group, const void* BW, const float* b1, float* ws, int64_t M, hipStream_t s) {
    using G = WgGeom<C>;
    int nchunk;
    const int rpc = rows_per_chunk_for<C>(M, &nchunk);
    static const bool raised = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&convnext_mlp_wgrad_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   G::LDS) == hipSuccess;
    }();
    (void)raised;
    float* part = ws;
    const int grid = 8 * G::NHG * ((nchunk + 7) / 8);
    hipLaunchKernelGGL((convnext_mlp_wgrad_kernel<C>), dim3(grid), dim3(G::NT), G::LDS, s, (const bf16_t*)y, mean, rstd, lng, lnb, (const bf16_t*)d,
                       rowscale, rows_per_group, BW, b1, part, M, (int64_t)rpc, nchunk);
    return iseg_check_launch("iseg_convnext_mlp_wgrad");
}

size_t wgrad_ws_floats(int C, int64_t M, int* nchunk_out) {
    int nchunk;
    if (C == 96) rows_per_chunk_for<96>(M, &nchunk);
    else rows_per_chunk_for<192>(M, &nchunk);
    if (nchunk_out) *nchunk_out = nchunk;
    const size_t rec = 2 * 4 * (size_t)C * C + 4 * C + C;      // [Z | dW1T | db1 | S]
    return (size_t)(nchunk + 1) * rec + (size_t)(4 * C / FIN_ROWS) * C;      // chunk records, their sum, the layer-scale partials
}

}  // namespace

extern "C" size_t iseg_convnext_mlp_wgrad_workspace_bytes(int64_t M, int C) {
    if (!(C == 96 || C == 192) || M <= 0) return 0;
    return wgrad_ws_floats(C, M, nullptr) * sizeof(float);
}

extern "C" int iseg_convnext_mlp_wgrad(const void* y, const float* mean, const float* rstd, const float* ln_gamma, const float* ln_beta,
                                       const void* dout, const float* rowscale, int64_t rows_per_group, const void* bw_tiled, const float* b1,
                                       const float* W2, const float* b2, const float* gamma, float* dW1, float* db1, float* dW2, float* db2,
                                       float* dgamma, int64_t M, int C, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(dtype == ISEG_BF16 && (C == 96 || C == 192), "iseg_convnext_mlp_wgrad: bf16 storage with C = 96 or 192 only (C = %d, dtype = %d)", C, dtype);
    ISEG_REQUIRE(y && dout && bw_tiled && b1 && W2 && b2 && dW1 && db1 && dW2 && db2 && M > 0, "iseg_convnext_mlp_wgrad: null operand or empty problem");
    ISEG_REQUIRE(!gamma || dgamma, "iseg_convnext_mlp_wgrad: gamma needs dgamma");
    ISEG_REQUIRE(!mean || (rstd && ln_gamma && ln_beta), "iseg_convnext_mlp_wgrad: mean needs rstd, ln_gamma and ln_beta");
    ISEG_REQUIRE(!rowscale || (rows_per_group > 0 && rows_per_group % 64 == 0),
                 "iseg_convnext_mlp_wgrad: the row factor must be constant inside 64-row tiles (rows_per_group = %lld)", (long long)rows_per_group);
    ISEG_REQUIRE((((uintptr_t)y | (uintptr_t)dout | (uintptr_t)bw_tiled | (uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0,
                 "iseg_convnext_mlp_wgrad: operands must be 16-byte aligned");
    int nchunk;
    const size_t need = wgrad_ws_floats(C, M, &nchunk) * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_convnext_mlp_wgrad: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    float* const wsf = (float*)ws;
    int rc = C == 96 ? launch_wgrad<96>(y, mean, rstd, ln_gamma, ln_beta, dout, rowscale, rows_per_group, bw_tiled, b1, wsf, M, stream)
                     : launch_wgrad<192>(y, mean, rstd, ln_gamma, ln_beta, dout, rowscale, rows_per_group, bw_tiled, b1, wsf, M, stream);
    if (rc != ISEG_OK) return rc;
    const int HID = 4 * C;
    const int64_t rec = 2 * (int64_t)HID * C + HID + C;
    float* part = wsf;
    float* sum = part + (int64_t)nchunk * rec;
    float* gpart_ws = sum + rec;
    // chunk records -> their sum: the fixed-order two-level row reduction every parameter gradient uses (common.h)
    launch_reduce_rows(part, nchunk, rec, 0, 1, rec, sum, nullptr, rec, 0, 1.f, 0, stream);
    const int P = HID / FIN_ROWS;
    float* gpart = gpart_ws;
    float* arena = nullptr;
    if (gamma) {
        arena = iseg_deferred_partials((size_t)P * C * sizeof(float), dgamma, nullptr, 1, stream);      // (common.h: deferred reductions)
        if (arena) gpart = arena;
    }
    hipLaunchKernelGGL(convnext_mlp_wgrad_finish_kernel, dim3(C / 32, HID / FIN_ROWS), dim3(256), 0, stream, sum, W2, b2, gamma,
                       dW1, db1, dW2, db2, gpart, C);
    if (gamma) {
        if (arena) iseg_deferred_push(gpart, P, C, C, dgamma, nullptr, C, 1.f, stream);
        else launch_reduce_rows(gpart, P, C, 0, 1, C, dgamma, nullptr, C, 0, 1.f, 1, stream);
    }
    return iseg_check_launch("iseg_convnext_mlp_wgrad_finish");
}
