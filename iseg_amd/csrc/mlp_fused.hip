// Fused ConvNeXt MLP (backbones/convnext.py:51-63 of the reference):
//     out = x + rowscale[sample] * gamma * (gelu(y2 @ W1 + b1) @ W2 + b2)
// for the wide, shallow stages (C = 96 / 192; M = 262144 / 65536 rows at the flagship size), where the [M, 4C] hidden tensor is
// 4x the activation and the un-fused pair of GEMMs is a pure HBM stream (two writes and two reads of 200 MB per block at stage 0).
// Here the hidden tile never leaves the CU in the forward pass, and the backward pass recomputes it.
//
// Everything is computed TRANSPOSED so that the first product's accumulator is directly the second product's MFMA operand
// (v_mfma_f32_32x32x16_bf16: the accumulator has its column on the lane and its rows in the 16 registers; a following MFMA that
// sums over those rows takes registers 8s..8s+7 as the B fragment of k-step s, no lane movement, no LDS):
//     H^T[hid][m] = W1^T[hid][c] . y2^T[c][m]        A = W1 slab, B = y2 rows (registers, loaded once)
//     G^T        = gelu(H^T + b1)                     in registers, rounded to bf16
//     O^T[c][m] += W2^T[c][hid] . G^T[hid][m]         A = W2 slab, B = G^T
// A wavefront owns 32 rows (m) and ALL C output channels (C/32 accumulator blocks).  The weights stream through a 3-stage LDS ring
// in slabs of 32 hidden units (global_load_lds_dwordx4, counted vmcnt, one raw s_barrier per stage -- the ring protocol of
// gemm_dma.h).
//
// Weight images.  A tiny per-step kernel (iseg_convnext_mlp_prep) rewrites the fp32 master kernels as bf16 *tiled* copies that are
// byte-for-byte the LDS images, slab after slab, so a ring stage is ONE contiguous run of 1-KiB pieces (piece p, lane l <- 16 bytes at
// p * 1024 + 16 l: perfectly coalesced DMA, no per-lane address math) and every A fragment is ONE plain ds_read_b128 of
// (row lane & 31, k-half lane >> 5) from a 1-KiB piece of 32 rows x 16 k.  Inside a piece the k-half is the OUTER index -- [2 k-halves][32 rows][8 k],
// so lane l reads bytes 16 l .. 16 l + 15: ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (and + 32), and
// only the lane-contiguous image puts each group on sixteen different 16-byte slots of the 256-byte bank row.  (Rounds 2-5 had the row outer,
// [32 rows][16 k]: 32-byte lane stride, every group on eight slots twice -- SQ_LDS_BANK_CONFLICT = 40 % of the LDS cycles, and with four SIMDs
// each asking for 1 KiB per 32-cycle MFMA a two-way conflict is exactly the LDS array's whole bandwidth.)  Pieces of a slab:
//     A1 (H = y2 W1):            [kk < C/16] pieces of (32 hid, 16 c)            value W1[16 kk + k][hid]
//     A2 (O = G W2):             [cb < C/32][s < 2] pieces of (32 c, 16 k*)      value W2[hid(s, k*)][32 cb + c]
//     A3 (dG = dbr (W2 gamma)^T):[kk < C/16] pieces of (32 hid, 16 c)            value W2[hid][16 kk + k] * gamma[16 kk + k]
//     A4 (dy2 = dH W1^T):        [cb < C/32][s < 2] pieces of (32 c, 16 k*)      value W1[32 cb + c][hid(s, k*)]
// k* is the accumulator's register order: position 8 h + j of k-step s is hidden unit 16 s + 8 (j >> 2) + 4 h + (j & 3) of the slab.
// (Plain C++ loads matter: hipcc orders a ds_read *builtin* behind every LDS-DMA in flight with s_waitcnt vmcnt(0) -- it carries no
// alias information -- which drained the ring on every stage in the first version of this file, built on ds_read_b64_tr_b16.)
// RING NOTE: inside the ring loop every LDS read must be a bf16x8 load like the fragments.  A float4-typed read of the bias copy made
// hipcc put s_waitcnt vmcnt(0) in front of it (it may alias the LDS-DMA as far as its alias info goes), draining the ring every
// stage; tools/check_mlp_isa.py greps the ISA for that.
// Forward buffer FW = [slab][A1 | A2] (C * 128 bytes per slab), backward buffer BW = [slab][A1 | A3 | A4] (C * 192 bytes).
//
// Per slab and wavefront: C/16 + C/16 MFMAs (32 cycles each) and 16 gelu evaluations per lane -> VALU-bound at C = 96, balanced at
// C = 192; two wavefronts per SIMD let one's gelu overlap the other's MFMAs.
#include "common.h"
#include "iseg_hip.h"
#include "mlp_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

constexpr int OUT_SLAB = 32 * 36 * 4;     // per-wavefront epilogue slab: 32 rows x (32 + 4 pad) floats

// NIMG images of C * 64 bytes per 32-hidden-unit slab, SUB slabs per ring stage, WAVES wavefronts of 32 rows.  Pieces of a stage are
// dealt round-robin over the wavefronts; when PIECES % WAVES != 0 the first NHI wavefronts issue one piece more, and each wavefront
// counts its own DMAs in the vmcnt wait.
template <int C, int SUB, int WAVES, int NIMG, int NSTAGES> struct MlpGeom {
    static constexpr int NS = NSTAGES;      // ring stages
    static constexpr int HID = 4 * C, KK = C / 16, CB = C / 32;
    static constexpr int IMG = C * 64;
    static constexpr int SLAB = NIMG * IMG;
    static constexpr int STAGE = SUB * SLAB;
    static constexpr int PIECES = STAGE / 1024;
    static constexpr int PLO = PIECES / WAVES, NHI = PIECES % WAVES, PHI = PLO + (NHI ? 1 : 0);
    static constexpr int NST = HID / (32 * SUB);
    static constexpr int RING = NS * STAGE > WAVES * OUT_SLAB ? NS * STAGE : WAVES * OUT_SLAB;
    static constexpr int LDS = RING + HID * 4;         // + b1 as floats
    static_assert(C % 32 == 0 && STAGE % 1024 == 0 && HID % (32 * SUB) == 0 && PLO >= 1 && NST >= NS - 1 && NS >= 2 && LDS <= 160 * 1024, "geometry");
};

__device__ __forceinline__ bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// One ring stage = PIECES consecutive KiB of the tiled weight buffer; wavefront `wid` moves pieces wid, wid + WAVES, ...
template <class G, int WAVES> struct RingFeeder {
    const char* src;      // this lane's 16 bytes of the wavefront's first piece of the next stage to issue
    int wid;
    __device__ __forceinline__ RingFeeder(const void* tiled, int wid_, int lane) : src((const char*)tiled + wid_ * 1024 + lane * 16), wid(wid_) {}
    __device__ __forceinline__ void issue(char* smem, int stage) {
        char* dst = smem + stage * G::STAGE + wid * 1024;
#pragma unroll
        for (int i = 0; i < G::PLO; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(src + i * WAVES * 1024), (lds_void_ptr)(dst + i * WAVES * 1024), 16, 0, 0);
        if (G::NHI && wid < G::NHI)
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(src + G::PLO * WAVES * 1024), (lds_void_ptr)(dst + G::PLO * WAVES * 1024), 16, 0, 0);
        src += G::STAGE;
    }
};


#define MLP_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

template <int C, int SUB, int WAVES, int NSTAGES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(C == 96 ? 4 : (C == 192 ? 2 : 1)))) void convnext_mlp_fwd_kernel(const bf16_t* __restrict__ Y, const void* __restrict__ FW,
                                                               const float* __restrict__ b1, const float* __restrict__ b2,
                                                               const float* __restrict__ gamma, const float* __restrict__ rowscale,
                                                               int64_t rows_per_group, const bf16_t* __restrict__ R, bf16_t* __restrict__ O,
                                                               int64_t M, MlpLayerNorm ln) {
    using G = MlpGeom<C, SUB, WAVES, 2, NSTAGES>;
    constexpr int HID = G::HID, KK = G::KK, CB = G::CB, IMG = G::IMG, SLAB = G::SLAB, STAGE = G::STAGE, NST = G::NST, NS = G::NS;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* const b1s = reinterpret_cast<float*>(smem + G::RING);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * (32 * WAVES) + wid * 32;

    // ---- B operand of the first product: this wavefront's 32 rows of y2, all C channels, straight from global memory ----
    bf16x8 yf[KK];
    {
        int64_t row = m0 + r;
        row = row < M ? row : M - 1;
        const bf16_t* yp = Y + row * C + 8 * h;
#ifdef ISEG_ABL_MLP_NOIO      // ablation (tools/kbench_mlp.py under ISEG_BUILD_DEFINES): no row loads, no residual loads, no stores
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int u = 0; u < 8; ++u) yf[kk][u] = (bf16_t)(0.01f * (float)((lane + kk + u) & 15));
        (void)yp;
#else
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) yf[kk] = *reinterpret_cast<const bf16x8*>(yp + 16 * kk);
#endif
        // Y is the LayerNorm INPUT (ln.gamma != NULL): a row lives in the two lanes r and r + 32, so the statistics are an in-lane sum and one
        // lane-half exchange; y2 never exists in HBM (round 3).  The row's mean / rstd are saved for the backward kernels.
        if (ln.gamma) mlp_layernorm_rows<KK>(yf, ln, m0 + r < M ? m0 + r : -1, h, C);
    }
    for (int i = tid; i < HID; i += 64 * WAVES) b1s[i] = b1[i];
    // the loads above must have landed before the ring starts (keeps the compiler's own vmcnt bookkeeping out of the loop)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" ::"v"(yf[kk]));

    RingFeeder<G, WAVES> feed(FW, wid, lane);

    f32x16 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[cb][j] = 0.f;

    const int fo = mlp_frag_offset(r, h);      // this lane's fragment inside a piece

    // A ring stage is one contiguous run of NF = SUB * (KK + 2 CB) fragment pieces (A1 then A2 of each slab) consumed in address order:
    // the fragments go through a rolling window of FD registers sets, FD pieces ahead of the MFMA that uses them.  Left to itself hipcc
    // sinks every ds_read_b128 next to its MFMA and waits for it (lgkmcnt(0) per MFMA: ~3x the MFMA time with one wavefront per SIMD);
    // the sched_barrier after each MFMA pins the order written here.
    constexpr int NF = SUB * (KK + 2 * CB), FD = C == 96 ? 3 : 4;      // (C = 96 runs two workgroups per CU: 128 registers, three fragments in flight is what fits)
    auto compute = [&](int stage, int kt) {
        const char* fb = smem + stage * STAGE + fo;
        bf16x8 win[FD];
#pragma unroll
        for (int d = 0; d < FD; ++d) win[d] = lds_frag(fb + d * 1024);
        auto take = [&](int f) {
            const bf16x8 a = win[f % FD];
            if (f + FD < NF) win[f % FD] = lds_frag(fb + (f + FD) * 1024);
            return a;
        };
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            const int f0 = sub * (KK + 2 * CB);
            f32x16 hacc;
            {
                const float* bb = b1s + 32 * (kt * SUB + sub) + 4 * h;      // register 4 i + u is hidden row 8 i + 4 h + u
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 v = __builtin_bit_cast(float4, lds_frag(reinterpret_cast<const char*>(bb + 8 * i)));      // (typed like the fragments: see RING NOTE)
                    hacc[4 * i] = v.x;
                    hacc[4 * i + 1] = v.y;
                    hacc[4 * i + 2] = v.z;
                    hacc[4 * i + 3] = v.w;
                }
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(take(f0 + kk), yf[kk], hacc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            bf16x8 gf[2];
#pragma unroll
            for (int j = 0; j < 16; ++j) gf[j >> 3][j & 7] = (bf16_t)gelu_poly(hacc[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(take(f0 + KK + 2 * cb + s), gf[s], acc[cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };

    // ---- ring: NS - 1 stages in flight, one barrier per stage (gemm_dma.h) ----
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) feed.issue(smem, p);
    int stage = 0, fill = NS - 1;
    for (int kt = 0; kt < NST; ++kt) {
        if (NST - 1 - kt >= NS - 2) {
            if (G::NHI && wid < G::NHI) MLP_WAIT((NS - 2) * G::PHI);
            else MLP_WAIT((NS - 2) * G::PLO);
        } else {
            MLP_WAIT(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < NST) feed.issue(smem, fill);
        compute(stage, kt);
        stage = stage + 1 == NS ? 0 : stage + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // the epilogue slabs overwrite the ring
    asm volatile("" ::: "memory");

    // ---- epilogue: O^T blocks -> per-wavefront LDS slab -> rows; lane l finishes 16 channels of row l >> 1 ----
    float* const slab = reinterpret_cast<float*>(smem + wid * OUT_SLAB);
    const int er = lane >> 1, eh = lane & 1;
    const int64_t m = m0 + er;
    float rs = 1.f;
    if (rowscale && m < M) rs = rowscale[m / rows_per_group];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(slab + r * 36 + 8 * i + 4 * h) = make_float4(acc[cb][4 * i], acc[cb][4 * i + 1], acc[cb][4 * i + 2], acc[cb][4 * i + 3]);
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(slab + er * 36 + 16 * eh + 4 * i);
            v[4 * i] = t.x;
            v[4 * i + 1] = t.y;
            v[4 * i + 2] = t.z;
            v[4 * i + 3] = t.w;
        }
        if (m < M) {
            const int c0 = 32 * cb + 16 * eh;
            float res[16], bb[16], gg[16];
#ifdef ISEG_ABL_MLP_NOIO
#pragma unroll
            for (int u = 0; u < 16; ++u) res[u] = 0.f;
#else
            load8<bf16_t>(R + m * C + c0, res);
            load8<bf16_t>(R + m * C + c0 + 8, res + 8);
#endif
            load8<float>(b2 + c0, bb);      // (16-byte vector loads: one dword load per element was 96 extra loads per lane at C = 96)
            load8<float>(b2 + c0 + 8, bb + 8);
            if (gamma) {
                load8<float>(gamma + c0, gg);
                load8<float>(gamma + c0 + 8, gg + 8);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                float t = v[u] + bb[u];
                if (gamma) t *= gg[u];
                v[u] = fmaf(t, rs, res[u]);
            }
#ifdef ISEG_ABL_MLP_NOIO
            if (v[0] == 12345.678f)
#endif
            {
                store8<bf16_t>(O + m * C + c0, v);
                store8<bf16_t>(O + m * C + c0 + 8, v + 8);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward chain of the same MLP.  Nothing [M, 4C]-shaped was saved by the forward pass: the pre-activation is recomputed from y2, and
//     H^T  = W1^T . y2^T + b1            G^T = gelu(H^T),  G'^T = gelu'(H^T)                       (as in the forward kernel)
//     dG^T = (W2 gamma)[hid][c] . dbr^T  A = A3, B = dbr rows (registers)
//     dH^T = dG^T o G'^T                 in registers, rounded to bf16 = B operand of
//     dy2^T[c][m] += W1[c][hid] . dH^T   A = A4
// G and dH leave the CU once, as bf16 [M, 4C] operands of the two weight-gradient GEMMs (Z = G^T dbr, dW1 = y2^T dH); dy2 [M, C] goes to
// the LayerNorm backward.  The accumulator layout (hidden units 8 i + 4 h + u in register 4 i + u of lane half h) is turned into 16-byte
// row pieces by one v_permlane32_swap per packed dword (lane halves exchange their odd / even groups).
// WAVES = 8 (C = 96, two wavefronts per SIMD share 256 registers each) or 4 (C = 192: y2 and dbr fragments + dy2 accumulators = 192
// registers, one wavefront per SIMD).
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (bf16_t)lo;
    v[1] = (bf16_t)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// accumulator registers (hidden unit 8 i + 4 h + u in v[4 i + u]) -> the lane's two 16-byte row pieces: hidden units 8 h .. 8 h + 7 and
// 16 + 8 h .. 16 + 8 h + 7 of row (lane & 31)
__device__ __forceinline__ void acc_to_row_pieces(const float* v, uint4& p0, uint4& p1) {
    uint32_t d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    // groups i = 0, 1 (dwords 0..3) and i = 2, 3 (dwords 4..7): swap the upper half's group k with the lower half's group k + 1
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            auto sw = __builtin_amdgcn_permlane32_swap(d[4 * pr + w], d[4 * pr + 2 + w], false, false);
            d[4 * pr + w] = sw[0];
            d[4 * pr + 2 + w] = sw[1];
        }
    p0 = make_uint4(d[0], d[1], d[2], d[3]);
    p1 = make_uint4(d[4], d[5], d[6], d[7]);
}

// STORE = false: the data-gradient chain only (dy2); G and dH never leave the CU -- the weight gradients come from mlp_wgrad.hip, which
// recomputes the hidden tile per hidden-unit slice.  `rowscale` (drop-path factor per group of rows_per_group rows) turns D = d(out) into
// dbr while the rows are loaded.
// LNB (with the LayerNorm on the row load, ln.gamma != NULL): the epilogue carries dy2 through the LayerNorm backward as well -- DY receives the
// gradient of the LayerNorm INPUT and ln_partials [gridDim.x][2 C] this workgroup's column sums (dgamma | dbeta) -- see the epilogue.
template <int C, int SUB, int WAVES, int NSTAGES, bool STORE, bool LNB = false>
__global__ __launch_bounds__(64 * WAVES) void convnext_mlp_bwd_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ D,
                                                                      const float* __restrict__ rowscale, int64_t rows_per_group,
                                                                      const void* __restrict__ BW, const float* __restrict__ b1,
                                                                      bf16_t* __restrict__ Gout, bf16_t* __restrict__ DHout,
                                                                      bf16_t* __restrict__ DY, int64_t M, MlpLayerNorm ln,
                                                                      float* __restrict__ ln_partials) {
    using G = MlpGeom<C, SUB, WAVES, 3, NSTAGES>;
    constexpr int HID = G::HID, KK = G::KK, CB = G::CB, IMG = G::IMG, SLAB = G::SLAB, STAGE = G::STAGE, NST = G::NST, NS = G::NS;
    constexpr int STORES = STORE ? 4 * SUB : 0;         // 16-byte global stores per wavefront and ring stage (G, dH: two each per slab)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* const b1s = reinterpret_cast<float*>(smem + G::RING);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * (32 * WAVES) + wid * 32;
    const bool active = m0 < M;             // wave-uniform: an inactive wavefront still feeds the ring and joins every barrier

    bf16x8 yf[KK], df[KK];
    {
        int64_t row = m0 + r;
        row = row < M ? row : M - 1;
        const bf16_t* yp = Y + row * C + 8 * h;
        const bf16_t* dp = D + row * C + 8 * h;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            yf[kk] = *reinterpret_cast<const bf16x8*>(yp + 16 * kk);
            df[kk] = *reinterpret_cast<const bf16x8*>(dp + 16 * kk);
        }
        if (ln.gamma) mlp_layernorm_apply_rows<KK>(yf, ln, row, h);      // Y = LayerNorm input, statistics saved by the forward kernel
        if (rowscale) {
            const float rs = rowscale[row / rows_per_group];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int u = 0; u < 8; ++u) df[kk][u] = (bf16_t)((float)df[kk][u] * rs);
        }
    }
    for (int i = tid; i < HID; i += 64 * WAVES) b1s[i] = b1[i];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" ::"v"(yf[kk]), "v"(df[kk]));

    RingFeeder<G, WAVES> feed(BW, wid, lane);

    f32x16 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[cb][j] = 0.f;

    const int fo = mlp_frag_offset(r, h);
    // this lane's 16-byte pieces of row m0 + r of G / dH: hidden units 8 h .. + 7 and 16 + 8 h .. + 7 of the current slab
    const bool row_ok = m0 + r < M;
    const int64_t row_o = (row_ok ? m0 + r : 0) * HID + 8 * h;

    // fragment pieces of a stage in address order: A1 (KK), A3 (KK), A4 (2 CB) of each slab -- rolling window as in the forward kernel
    constexpr int NF = SUB * (2 * KK + 2 * CB), FD = C >= 384 ? 2 : 4;      // (C = 384 runs at 490 of 512 registers)
    auto compute = [&](int stage, int kt) {
        const char* fb = smem + stage * STAGE + fo;
        bf16x8 win[FD];
#pragma unroll
        for (int d = 0; d < FD; ++d) win[d] = lds_frag(fb + d * 1024);
        auto take = [&](int f) {
            const bf16x8 a = win[f % FD];
            if (f + FD < NF) win[f % FD] = lds_frag(fb + (f + FD) * 1024);
            return a;
        };
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            const int f0 = sub * (2 * KK + 2 * CB);
            f32x16 hacc, dacc;
            {
                const float* bb = b1s + 32 * (kt * SUB + sub) + 4 * h;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 v = __builtin_bit_cast(float4, lds_frag(reinterpret_cast<const char*>(bb + 8 * i)));      // (typed like the fragments: see RING NOTE)
                    hacc[4 * i] = v.x;
                    hacc[4 * i + 1] = v.y;
                    hacc[4 * i + 2] = v.z;
                    hacc[4 * i + 3] = v.w;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) dacc[j] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(take(f0 + kk), yf[kk], hacc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(take(f0 + KK + kk), df[kk], dacc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            float gv[16], dv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float gd;
                if constexpr (STORE) gelu_sig_both(hacc[j], gv[j], gd);
                else gd = gelu_poly_grad(hacc[j]);      // the chain alone needs the derivative only: 12 full-rate instructions, no transcendental
                dv[j] = dacc[j] * gd;
            }
            bf16x8 hf[2];
#pragma unroll
            for (int j = 0; j < 16; ++j) hf[j >> 3][j & 7] = (bf16_t)dv[j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(take(f0 + 2 * KK + 2 * cb + s), hf[s], acc[cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (STORE) {
                uint4 g0, g1, d0, d1;
                acc_to_row_pieces(gv, g0, g1);
                acc_to_row_pieces(dv, d0, d1);
                if (row_ok) {
                    const int64_t o = row_o + 32 * (kt * SUB + sub);
                    *reinterpret_cast<uint4*>(Gout + o) = g0;
                    *reinterpret_cast<uint4*>(Gout + o + 16) = g1;
                    *reinterpret_cast<uint4*>(DHout + o) = d0;
                    *reinterpret_cast<uint4*>(DHout + o + 16) = d1;
                }
            }
        }
    };

#pragma unroll
    for (int p = 0; p < NS - 1; ++p) feed.issue(smem, p);
    int stage = 0, fill = NS - 1;
    for (int kt = 0; kt < NST; ++kt) {
        // Stage kt must have landed.  vmcnt retires in issue order, so the wait allows exactly the operations issued AFTER stage kt's DMAs:
        // the G / dH stores of iteration kt-2 (issued behind DMA(kt)), the DMAs of the NS-2 following stages, the stores of iteration kt-1.
        // (Allowing fewer -- the first version insisted on the kt-2 stores -- stalls every stage on the write acknowledgements of a kernel
        // that streams 200 MB out; allowing more would let pieces of stage kt itself be pending.)
        {
            const bool more = NST - 1 - kt >= NS - 2;      // the following stages' DMAs are in flight
            const bool hi = G::NHI && wid < G::NHI;
#define MLP_WAIT_BWD(S_)                                                     \
    do {                                                                     \
        if (more) {                                                          \
            if (hi) MLP_WAIT((NS - 2) * G::PHI + (S_));                      \
            else MLP_WAIT((NS - 2) * G::PLO + (S_));                         \
        } else {                                                             \
            MLP_WAIT(S_);                                                    \
        }                                                                    \
    } while (0)
            if (!active || kt == 0) MLP_WAIT_BWD(0);
            else if (kt == 1) MLP_WAIT_BWD(STORES);
            else MLP_WAIT_BWD(2 * STORES);
#undef MLP_WAIT_BWD
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < NST) feed.issue(smem, fill);
        if (active) compute(stage, kt);
        stage = stage + 1 == NS ? 0 : stage + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- dy2^T blocks -> per-wavefront LDS slab -> bf16 rows ----
    float* const slab = reinterpret_cast<float*>(smem + wid * OUT_SLAB);
    const int er = lane >> 1, eh = lane & 1;
    const int64_t m = m0 + er;
    // lane (er, eh) <- columns 32 cb + 16 eh .. + 15 of row er of the wavefront's dy2 block (the slab is wave-private; LDS operations of a
    // wavefront complete in order)
    auto exchange = [&](int cb, float* v) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(slab + r * 36 + 8 * i + 4 * h) = make_float4(acc[cb][4 * i], acc[cb][4 * i + 1], acc[cb][4 * i + 2], acc[cb][4 * i + 3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(slab + er * 36 + 16 * eh + 4 * i);
            v[4 * i] = t.x;
            v[4 * i + 1] = t.y;
            v[4 * i + 2] = t.z;
            v[4 * i + 3] = t.w;
        }
    };
    if constexpr (!LNB) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            float v[16];
            exchange(cb, v);
            if (m < M) {
                const int c0 = 32 * cb + 16 * eh;
                store8<bf16_t>(DY + m * C + c0, v);
                store8<bf16_t>(DY + m * C + c0 + 8, v + 8);
            }
        }
    } else {
        // LayerNorm backward on the rows while they are here (keras LayerNormalization, the arithmetic of layernorm_bwd_kernel in norm.hip):
        //   t = dy2 * gamma,  dy1 = rstd * (t - mean_c(t) - xhat * mean_c(t * xhat)),  dgamma += dy2 * xhat,  dbeta += dy2   (column sums)
        // Two passes over the column blocks (the accumulators stay in registers, so the exchange is simply repeated): row sums first, then the
        // outputs; the column sums go through the same slab -- a lane adds up 16 rows of one column, halves meet by one swap, wavefronts in
        // wavefront order -- and leave as ONE partial row per workgroup for the fixed-order reduction: no atomics, bit-reproducible (the
        // LayerNorm backward kernel this replaces combines its lanes by LDS float atomics).
        // (every global operand of the epilogue is requested up front -- the row's LayerNorm input as raw bf16, its statistics, gamma through
        // LDS: loaded where they are used, the 2 x CB dependent round trips cost 26 us per launch, more than the LayerNorm kernel they replace)
        float* const lnacc = reinterpret_cast<float*>(smem + WAVES * OUT_SLAB);      // [WAVES][2][C]: inside the ring, which is free by now
        float* const gam_s = lnacc + WAVES * 2 * C;                                  // [C]
        const int64_t mr = m < M ? m : M - 1;
        bf16x8 yraw[CB][2];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            yraw[cb][0] = *reinterpret_cast<const bf16x8*>(Y + mr * C + 32 * cb + 16 * eh);
            yraw[cb][1] = *reinterpret_cast<const bf16x8*>(Y + mr * C + 32 * cb + 16 * eh + 8);
        }
        const float mean = ln.mean[mr], rstd = ln.rstd[mr];
        for (int i = tid; i < WAVES * 2 * C; i += 64 * WAVES) lnacc[i] = 0.f;      // (an inactive wavefront's row stays zero)
        for (int i = tid; i < C; i += 64 * WAVES) gam_s[i] = ln.gamma[i];
        __syncthreads();
        const bool live = active && m < M;
        float s1 = 0.f, s2 = 0.f;
        auto xhat16 = [&](int cb, float* xh) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xh[u] = ((float)yraw[cb][0][u] - mean) * rstd;
                xh[8 + u] = ((float)yraw[cb][1][u] - mean) * rstd;
            }
        };
        auto gamma16 = [&](int c0, float* gm) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(gam_s + c0 + 4 * i);
                gm[4 * i] = t.x, gm[4 * i + 1] = t.y, gm[4 * i + 2] = t.z, gm[4 * i + 3] = t.w;
            }
        };
        if (active) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int c0 = 32 * cb + 16 * eh;
                float v[16], xh[16], gm[16];
                exchange(cb, v);
                xhat16(cb, xh);
                gamma16(c0, gm);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float t = v[u] * gm[u];
                    s1 += t;
                    s2 = fmaf(t, xh[u], s2);
                }
            }
            s1 += __shfl_xor(s1, 1);      // the two lanes of a row
            s2 += __shfl_xor(s2, 1);
            s1 *= 1.f / (float)C;
            s2 *= 1.f / (float)C;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int c0 = 32 * cb + 16 * eh;
                float v[16], xh[16], gm[16], o[16];
                exchange(cb, v);
                xhat16(cb, xh);
                gamma16(c0, gm);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (!live) v[u] = 0.f;      // rows past M add nothing to the column sums
                    o[u] = rstd * (v[u] * gm[u] - s1 - xh[u] * s2);
                }
                if (live) {
                    store8<bf16_t>(DY + m * C + c0, o);
                    store8<bf16_t>(DY + m * C + c0 + 8, o + 8);
                }
                // column sums of dy2 * xhat (q = 0) and dy2 (q = 1) over the wavefront's 32 rows
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<float4*>(slab + er * 36 + 16 * eh + 4 * i) =
                            q == 0 ? make_float4(v[4 * i] * xh[4 * i], v[4 * i + 1] * xh[4 * i + 1], v[4 * i + 2] * xh[4 * i + 2], v[4 * i + 3] * xh[4 * i + 3])
                                   : make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
                    float cs = 0.f;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) cs += slab[(16 * h + rr) * 36 + r];      // lane (r = column, h = row half)
                    cs += __shfl_xor(cs, 32);
                    if (h == 0) lnacc[(wid * 2 + q) * C + 32 * cb + r] = cs;      // this wavefront's row: no atomics, the sum below is ordered
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * C; i += 64 * WAVES) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) t += lnacc[wv * 2 * C + i];      // wavefront order: bit-reproducible
            ln_partials[(int64_t)blockIdx.x * 2 * C + i] = t;
        }
    }
}

// fp32 master kernels -> the tiled bf16 images (see the header comment).  One thread per image element.
__device__ __forceinline__ void mlp_prep_elements(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ gamma,
                                                  bf16_t* __restrict__ FW, bf16_t* __restrict__ BW, int C, int first, int stride) {
    const int HID = 4 * C, per_img = C * 32, nslab = HID / 32;
    const int nfw = nslab * 2 * per_img, nbw = BW ? nslab * 3 * per_img : 0;
    for (int e = first; e < nfw + nbw; e += stride) {
        const bool bw = e >= nfw;
        const int i = bw ? e - nfw : e, nimg = bw ? 3 : 2;
        const int slab = i / (nimg * per_img), rem = i % (nimg * per_img);
        const int img = rem / per_img, x = rem % per_img;
#ifdef ISEG_MLP_PIECE_ROWMAJOR      // (the piece layout of rounds 2-5, kept for A/B builds: [32 rows][16 k], 2-way bank conflicts on every fragment read)
        const int piece = x / 512, y = x % 512, rr = y / 16, kq = y % 16, hh = kq / 8, j = kq % 8;
#else                               // [k-half][32 rows][8 k]: lane l = 32 (k-half) + row reads the piece's bytes 16 l .. 16 l + 15
        const int piece = x / 512, y = x % 512, hh = y / 256, rr = (y % 256) / 8, j = y % 8, kq = 8 * hh + j;
#endif
        // image kinds: 0 = A1, 1 = A2 (forward) ; 0 = A1, 1 = A3, 2 = A4 (backward)
        const int kind = bw ? (img == 0 ? 1 : img + 2) : img + 1;      // 1 = A1, 2 = A2, 3 = A3, 4 = A4
        float v;
        if (kind == 1 || kind == 3) {
            const int hid = 32 * slab + rr, c = 16 * piece + kq;
            v = kind == 1 ? W1[(int64_t)c * HID + hid] : W2[(int64_t)hid * C + c] * (gamma ? gamma[c] : 1.f);
        } else {
            const int cb = piece / 2, s = piece % 2;
            const int c = 32 * cb + rr, hid = 32 * slab + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
            v = kind == 2 ? W2[(int64_t)hid * C + c] : W1[(int64_t)c * HID + hid];
        }
        (bw ? BW : FW)[i] = (bf16_t)v;
    }
}

__global__ void convnext_mlp_prep_kernel(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ gamma,
                                         bf16_t* __restrict__ FW, bf16_t* __restrict__ BW, int C) {
    mlp_prep_elements(W1, W2, gamma, FW, BW, C, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// Every per-weight-update derivation of the ConvNeXt blocks in ONE launch (blockIdx.y = table entry): the tiled MLP images of the fused
// stages (kind 1) and the layer-scale-folded bf16 kernels W2 * gamma of the un-fused stages' data gradients (kind 0) -- 18 launches of ~5 us
// per step before.  Entry = 8 x int64: {kind, W1 | src, W2, gamma, FW | dst, BW, C | cols, rows}.
__global__ void convnext_weight_prep_batched_kernel(const int64_t* __restrict__ table) {
    const int64_t* e = table + 8 * blockIdx.y;
    const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (e[0] == 1) {
        mlp_prep_elements(reinterpret_cast<const float*>(e[1]), reinterpret_cast<const float*>(e[2]), reinterpret_cast<const float*>(e[3]),
                          reinterpret_cast<bf16_t*>(e[4]), reinterpret_cast<bf16_t*>(e[5]), (int)e[6], first, stride);
    } else if (e[0] == 2) {
        // kind 2: two kernels on the same input, [C][Na] and [C][Nb] (+ their biases), as the images of ONE product of width ld >= Na + Nb (zero
        // columns behind): dst = { [C][ld] bf16 (the data gradient's operand) | [ld][C] bf16 (the forward product's K-contiguous operand) |
        // [ld] fp32 bias }.  Entry: {2, Wa, Wb, ba, dst, bb, C, Na | Nb << 20 | ld << 40}   (the DCNv3 offset | mask projection, csrc/dcnv3.hip)
        const float* wa = reinterpret_cast<const float*>(e[1]);
        const float* wb = reinterpret_cast<const float*>(e[2]);
        const float* ba = reinterpret_cast<const float*>(e[3]);
        const float* bb = reinterpret_cast<const float*>(e[5]);
        const int Cc = (int)e[6], na = (int)(e[7] & 0xFFFFF), nb = (int)((e[7] >> 20) & 0xFFFFF), ld = (int)(e[7] >> 40);
        bf16_t* rowcat = reinterpret_cast<bf16_t*>(e[4]);
        bf16_t* tr = rowcat + (int64_t)Cc * ld;
        float* bias = reinterpret_cast<float*>(tr + (int64_t)Cc * ld);
        const int64_t total = (int64_t)Cc * ld;
        for (int64_t i = first; i < total; i += stride) {
            const int c = (int)(i / ld), j = (int)(i % ld);
            const float v = j < na ? wa[(int64_t)c * na + j] : (j < na + nb ? wb[(int64_t)c * nb + (j - na)] : 0.f);
            rowcat[i] = (bf16_t)v;
            tr[(int64_t)j * Cc + c] = (bf16_t)v;
        }
        for (int j = first; j < ld; j += stride) bias[j] = j < na ? (ba ? ba[j] : 0.f) : (j < na + nb ? (bb ? bb[j - na] : 0.f) : 0.f);
    } else {
        const float* src = reinterpret_cast<const float*>(e[1]);
        const float* g = reinterpret_cast<const float*>(e[3]);
        bf16_t* dst = reinterpret_cast<bf16_t*>(e[4]);
        const int cols = (int)e[6];
        const int64_t total = e[7] * cols;
        for (int64_t i = first; i < total; i += stride) dst[i] = (bf16_t)(src[i] * g[i % cols]);
    }
}

template <int C, int SUB, int WAVES, int NSTAGES>
int launch_mlp_fwd(const void* y2, const void* FW, const float* b1, const float* b2, const float* gamma, const float* rowscale,
                   int64_t rows_per_group, const void* residual, void* out, int64_t M, const MlpLayerNorm& ln, hipStream_t s) {
    using G = MlpGeom<C, SUB, WAVES, 2, NSTAGES>;
    static const bool raised = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&convnext_mlp_fwd_kernel<C, SUB, WAVES, NSTAGES>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) == hipSuccess;
    }();
    (void)raised;
    const int grid = (int)ceil_div64(M, 32 * WAVES);
    hipLaunchKernelGGL((convnext_mlp_fwd_kernel<C, SUB, WAVES, NSTAGES>), dim3(grid), dim3(64 * WAVES), G::LDS, s, (const bf16_t*)y2, FW, b1, b2, gamma, rowscale,
                       rows_per_group, (const bf16_t*)residual, (bf16_t*)out, M, ln);
    return iseg_check_launch("iseg_convnext_mlp_fwd");
}

template <int C, int SUB, int WAVES, int NSTAGES, bool STORE, bool LNB = false>
int launch_mlp_bwd(const void* y2, const void* dbr, const float* rowscale, int64_t rows_per_group, const void* BW, const float* b1, void* g,
                   void* dh, void* dy2, int64_t M, const MlpLayerNorm& ln, hipStream_t s, float* ln_partials = nullptr) {
    using G = MlpGeom<C, SUB, WAVES, 3, NSTAGES>;
    static_assert(!LNB || G::RING >= WAVES * OUT_SLAB + (WAVES * 2 + 1) * C * 4, "the column sums and gamma sit behind the epilogue slabs inside the ring");
    static const bool raised = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&convnext_mlp_bwd_kernel<C, SUB, WAVES, NSTAGES, STORE, LNB>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) == hipSuccess;
    }();
    (void)raised;
    const int grid = (int)ceil_div64(M, 32 * WAVES);
    hipLaunchKernelGGL((convnext_mlp_bwd_kernel<C, SUB, WAVES, NSTAGES, STORE, LNB>), dim3(grid), dim3(64 * WAVES), G::LDS, s, (const bf16_t*)y2,
                       (const bf16_t*)dbr, rowscale, rows_per_group, BW, b1, (bf16_t*)g, (bf16_t*)dh, (bf16_t*)dy2, M, ln, ln_partials);
    return iseg_check_launch("iseg_convnext_mlp_bwd");
}

}  // namespace

extern "C" int iseg_convnext_mlp_supported(int C, int dtype) {
    // C = 384 is instantiated (one wavefront per SIMD, 128-row workgroups, a two-stage ring in the backward kernel) but measured SLOWER than the
    // GEMM pair at the flagship's 16384 rows: 115 vs 103 us forward, 199 vs 69 + 36 us backward -- only 128 workgroups, and with one wavefront
    // per SIMD nothing overlaps the gelu VALU section with MFMAs.  Opt-in for experiments: ISEG_MLP_FUSED_384=1.
    static const bool wide = [] { const char* e = getenv("ISEG_MLP_FUSED_384"); return e && atoi(e) != 0; }();
    return dtype == ISEG_BF16 && (C == 96 || C == 192 || (C == 384 && wide)) ? 1 : 0;
}

extern "C" size_t iseg_convnext_mlp_tiled_bytes(int C, int backward) { return (size_t)(backward ? 3 : 2) * 4 * C * C * 2; }

extern "C" int iseg_convnext_mlp_prep(const float* W1, const float* W2, const float* gamma, void* fw_tiled, void* bw_tiled, int C,
                                      hipStream_t stream) {
    ISEG_REQUIRE(W1 && W2 && fw_tiled && C > 0 && C % 32 == 0, "iseg_convnext_mlp_prep: bad arguments");
    const int total = (bw_tiled ? 5 : 2) * 4 * C * C;
    hipLaunchKernelGGL(convnext_mlp_prep_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, W1, W2, gamma, (bf16_t*)fw_tiled,
                       (bf16_t*)bw_tiled, C);
    return iseg_check_launch("iseg_convnext_mlp_prep");
}

extern "C" int iseg_convnext_weight_prep_batched(const int64_t* table, int entries, int64_t max_elements, hipStream_t stream) {
    ISEG_REQUIRE(table && entries > 0 && max_elements > 0, "iseg_convnext_weight_prep_batched: bad arguments");
    int64_t bx = (max_elements + 255) / 256;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(convnext_weight_prep_batched_kernel, dim3((unsigned)bx, (unsigned)entries), dim3(256), 0, stream, table);
    return iseg_check_launch("iseg_convnext_weight_prep_batched");
}

extern "C" int iseg_convnext_mlp_fwd(const void* y2, const void* fw_tiled, const float* b1, const float* b2, const float* gamma,
                                     const float* rowscale, int64_t rows_per_group, const void* residual, void* out, int64_t M, int C, int dtype,
                                     hipStream_t stream) {
    ISEG_REQUIRE(iseg_convnext_mlp_supported(C, dtype), "iseg_convnext_mlp_fwd: bf16 storage with C = 96, 192 or 384 only (C = %d, dtype = %d)", C, dtype);
    ISEG_REQUIRE(y2 && fw_tiled && b1 && b2 && residual && out && M > 0, "iseg_convnext_mlp_fwd: null operand or empty problem");
    ISEG_REQUIRE(!rowscale || rows_per_group > 0, "iseg_convnext_mlp_fwd: rowscale needs rows_per_group > 0");
    ISEG_REQUIRE((((uintptr_t)y2 | (uintptr_t)fw_tiled | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)gamma) & 15) == 0,
                 "iseg_convnext_mlp_fwd: operands must be 16-byte aligned");
    const MlpLayerNorm none{};
    if (C == 96) return launch_mlp_fwd<96, 2, 8, 3>(y2, fw_tiled, b1, b2, gamma, rowscale, rows_per_group, residual, out, M, none, stream);
    if (C == 192) return launch_mlp_fwd<192, 1, 8, 3>(y2, fw_tiled, b1, b2, gamma, rowscale, rows_per_group, residual, out, M, none, stream);
    // C = 384: 96 fragment + 192 accumulator registers per lane -> one wavefront per SIMD, 128-row workgroups
    return launch_mlp_fwd<384, 1, 4, 3>(y2, fw_tiled, b1, b2, gamma, rowscale, rows_per_group, residual, out, M, none, stream);
}

extern "C" int iseg_convnext_mlp_fwd_ln(const void* y1, const float* ln_gamma, const float* ln_beta, float eps, float* mean, float* rstd,
                                        const void* fw_tiled, const float* b1, const float* b2, const float* gamma, const float* rowscale,
                                        int64_t rows_per_group, const void* residual, void* out, int64_t M, int C, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(dtype == ISEG_BF16 && (C == 96 || C == 192), "iseg_convnext_mlp_fwd_ln: bf16 storage with C = 96 or 192 only (C = %d, dtype = %d)", C, dtype);
    ISEG_REQUIRE(y1 && ln_gamma && ln_beta && mean && rstd && fw_tiled && b1 && b2 && residual && out && M > 0,
                 "iseg_convnext_mlp_fwd_ln: null operand or empty problem");
    ISEG_REQUIRE(!rowscale || rows_per_group > 0, "iseg_convnext_mlp_fwd_ln: rowscale needs rows_per_group > 0");
    ISEG_REQUIRE((((uintptr_t)y1 | (uintptr_t)fw_tiled | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)gamma |
                   (uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0, "iseg_convnext_mlp_fwd_ln: operands must be 16-byte aligned");
    const MlpLayerNorm ln{ln_gamma, ln_beta, mean, rstd, eps};
    if (C == 96) return launch_mlp_fwd<96, 2, 8, 3>(y1, fw_tiled, b1, b2, gamma, rowscale, rows_per_group, residual, out, M, ln, stream);
    return launch_mlp_fwd<192, 1, 8, 3>(y1, fw_tiled, b1, b2, gamma, rowscale, rows_per_group, residual, out, M, ln, stream);
}

extern "C" int iseg_convnext_mlp_bwd(const void* y2, const void* dbr, const void* bw_tiled, const float* b1, void* g, void* dh, void* dy2,
                                     int64_t M, int C, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(iseg_convnext_mlp_supported(C, dtype), "iseg_convnext_mlp_bwd: bf16 storage with C = 96, 192 or 384 only (C = %d, dtype = %d)", C, dtype);
    ISEG_REQUIRE(y2 && dbr && bw_tiled && b1 && g && dh && dy2 && M > 0, "iseg_convnext_mlp_bwd: null operand or empty problem");
    ISEG_REQUIRE((((uintptr_t)y2 | (uintptr_t)dbr | (uintptr_t)bw_tiled | (uintptr_t)b1 | (uintptr_t)g | (uintptr_t)dh | (uintptr_t)dy2) & 15) == 0,
                 "iseg_convnext_mlp_bwd: operands must be 16-byte aligned");
    const MlpLayerNorm none{};
    if (C == 96) return launch_mlp_bwd<96, 2, 8, 3, true>(y2, dbr, nullptr, 0, bw_tiled, b1, g, dh, dy2, M, none, stream);
    if (C == 192) return launch_mlp_bwd<192, 1, 4, 3, true>(y2, dbr, nullptr, 0, bw_tiled, b1, g, dh, dy2, M, none, stream);
    return launch_mlp_bwd<384, 1, 4, 2, true>(y2, dbr, nullptr, 0, bw_tiled, b1, g, dh, dy2, M, none, stream);      // 72-KiB slabs: a two-stage ring is what fits
}

extern "C" int iseg_convnext_mlp_bwd_data(const void* y2, const float* mean, const float* rstd, const float* ln_gamma, const float* ln_beta,
                                          const void* dout, const float* rowscale, int64_t rows_per_group, const void* bw_tiled,
                                          const float* b1, void* dy2, int64_t M, int C, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(dtype == ISEG_BF16 && (C == 96 || C == 192), "iseg_convnext_mlp_bwd_data: bf16 storage with C = 96 or 192 only (C = %d, dtype = %d)", C, dtype);
    ISEG_REQUIRE(y2 && dout && bw_tiled && b1 && dy2 && M > 0, "iseg_convnext_mlp_bwd_data: null operand or empty problem");
    ISEG_REQUIRE(!rowscale || rows_per_group > 0, "iseg_convnext_mlp_bwd_data: rowscale needs rows_per_group > 0");
    ISEG_REQUIRE((((uintptr_t)y2 | (uintptr_t)dout | (uintptr_t)bw_tiled | (uintptr_t)b1 | (uintptr_t)dy2) & 15) == 0,
                 "iseg_convnext_mlp_bwd_data: operands must be 16-byte aligned");
    ISEG_REQUIRE(!mean || (rstd && ln_gamma && ln_beta && (((uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0),
                 "iseg_convnext_mlp_bwd_data: mean needs rstd and 16-byte aligned ln_gamma / ln_beta");
    const MlpLayerNorm ln{mean ? ln_gamma : nullptr, ln_beta, const_cast<float*>(mean), const_cast<float*>(rstd), 0.f};
    if (C == 96) return launch_mlp_bwd<96, 2, 8, 3, false>(y2, dout, rowscale, rows_per_group, bw_tiled, b1, nullptr, nullptr, dy2, M, ln, stream);
    return launch_mlp_bwd<192, 1, 4, 3, false>(y2, dout, rowscale, rows_per_group, bw_tiled, b1, nullptr, nullptr, dy2, M, ln, stream);
}

static int64_t mlp_bwd_blocks(int64_t M, int C) { return ceil_div64(M, 32 * (C == 96 ? 8 : 4)); }

extern "C" size_t iseg_convnext_mlp_bwd_data_ln_workspace_bytes(int64_t M, int C) {
    return (C == 96 || C == 192) && M > 0 ? (size_t)mlp_bwd_blocks(M, C) * 2 * C * sizeof(float) : 0;
}

extern "C" int iseg_convnext_mlp_bwd_data_ln(const void* y1, const float* mean, const float* rstd, const float* ln_gamma, const float* ln_beta,
                                             const void* dout, const float* rowscale, int64_t rows_per_group, const void* bw_tiled,
                                             const float* b1, void* dy1, float* dln_gamma, float* dln_beta, int64_t M, int C, int dtype,
                                             void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(dtype == ISEG_BF16 && (C == 96 || C == 192), "iseg_convnext_mlp_bwd_data_ln: bf16 storage with C = 96 or 192 only (C = %d, dtype = %d)", C, dtype);
    ISEG_REQUIRE(y1 && mean && rstd && ln_gamma && ln_beta && dout && bw_tiled && b1 && dy1 && dln_gamma && dln_beta && M > 0,
                 "iseg_convnext_mlp_bwd_data_ln: null operand or empty problem");
    ISEG_REQUIRE(!rowscale || rows_per_group > 0, "iseg_convnext_mlp_bwd_data_ln: rowscale needs rows_per_group > 0");
    ISEG_REQUIRE((((uintptr_t)y1 | (uintptr_t)dout | (uintptr_t)bw_tiled | (uintptr_t)b1 | (uintptr_t)dy1 | (uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0,
                 "iseg_convnext_mlp_bwd_data_ln: operands must be 16-byte aligned");
    const size_t need = iseg_convnext_mlp_bwd_data_ln_workspace_bytes(M, C);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_convnext_mlp_bwd_data_ln: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const MlpLayerNorm ln{ln_gamma, ln_beta, const_cast<float*>(mean), const_cast<float*>(rstd), 0.f};
    // dgamma | dbeta: the workgroups' partial rows, summed in fixed order into the gradient buffers (queued when the caller defers reductions)
    float* partials = (float*)ws;
    float* const arena = iseg_deferred_partials(need, dln_gamma, dln_beta, 1, stream);
    if (arena) partials = arena;
    const int rc = C == 96 ? launch_mlp_bwd<96, 2, 8, 3, false, true>(y1, dout, rowscale, rows_per_group, bw_tiled, b1, nullptr, nullptr, dy1, M, ln,
                                                                       stream, partials)
                           : launch_mlp_bwd<192, 1, 4, 3, false, true>(y1, dout, rowscale, rows_per_group, bw_tiled, b1, nullptr, nullptr, dy1, M, ln,
                                                                        stream, partials);
    if (rc != ISEG_OK) return rc;
    const int blocks = (int)mlp_bwd_blocks(M, C);
    if (arena) iseg_deferred_push(partials, blocks, 2 * C, 2 * C, dln_gamma, dln_beta, C, 1.f, stream);
    else launch_reduce_rows(partials, blocks, 2 * C, 0, 1, 2 * C, dln_gamma, dln_beta, C, 0, 1.f, 1, stream);
    return iseg_check_launch("iseg_convnext_mlp_bwd_data_ln");
}
