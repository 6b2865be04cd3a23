// Depthwise 7 x 7 convolution (stride 1, dilation 1, bf16 storage, C % 32 == 0) on the matrix cores -- round 5.
// keras.layers.DepthwiseConv2D of the ConvNeXt block (backbones/convnext.py:23-27, 47-50) and its data gradient (the same product with the
// kernel flipped).  BASELINE configs[1] names it: "implicit-GEMM 7x7 depthwise".
//
// A depthwise convolution has one kernel per channel, so the only dense contraction in it is the banded (Toeplitz) product along one image axis:
//     out[r][c] = sum_kx sum_rho  X[rho][c + kx] * T_kx[rho][r],      T_kx[rho][r] = w[rho - r][kx]  (0 <= rho - r <= 6, else 0)
// per channel.  v_mfma_f32_4x4x4_16b_bf16 computes SIXTEEN independent 4 x 4 x 4 products per instruction (block b = lanes 4b .. 4b+3): block =
// channel.  Per channel block and kernel column kx:
//     A_b[m][k] = X[4 chunk + k][c0 + m + kx]     m = 4 output columns, k = 4 input rows of row chunk `chunk`   (lane 4b + m: 8 bytes)
//     B_b[k][n] = w[4 q + k - n][kx]              n = 4 output rows, q = chunk - row block in {0, 1, 2}           (lane 4b + n: 8 bytes)
//     D_b[m][n] += A_b B_b                        lane 4b + n holds output row 4 j + n, columns c0 .. c0 + 3 in its four registers
// so a 4 x 4 output patch of 16 channels costs 7 x 3 = 21 MFMAs (8 cycles each; 7 of 12 multiplied rows carry a tap), and
//   * the 21 Toeplitz fragments of a wavefront's 16 channels are weights only: 42 registers, built once per channel slab;
//   * a data fragment (one kx, one row chunk) serves the three row blocks it overlaps: walking down a 4-column strip costs SEVEN 8-byte LDS reads
//     per 21 MFMAs -- 14 bytes of LDS per output element against 28 for the 16 x 16 x 32 form (and that one needs a 1-KiB Toeplitz fragment per
//     MFMA on top: the banded operand is per channel, it cannot be shared across the M rows).
// The fragments want the image planar ([channel][column][row], rows contiguous); HBM holds NHWC.  The tile therefore goes
//   HBM --global_load_lds--> raw NHWC tile [24 rows][23 columns][32 channels]  (zero page for padding, as dwconv.hip)
//       --ds_read_b64_tr_b16--> registers: lane = channel, 4 consecutive rows  (the transposing read: block row = image row, block columns = 16
//                               channels; a 32-lane half takes both 16-channel halves of one pixel column: conflict-free with the 23-pixel pitch)
//       --ds_write_b64--> planar image P[32][22 columns][28 rows] (plane 1248 B, column 56 B: the fragment reads of a half-wave -- 8 channels x
//                               4 columns x 8 bytes -- tile the 64 banks exactly)
// and the outputs go accumulators -> fp32 NHWC tile in LDS (over the dead raw tile) -> 16-byte bf16 pieces (+ the residual-branch gradient) -> HBM,
// rounded once.  Weights are rounded to bf16 (the reference's mixed_bfloat16 policy casts the depthwise kernel to the compute dtype,
// utils/common.py:32-64); products and sums are fp32 in the matrix pipe.
//
// Unit = (32-channel slab, image, 16 x 16 output tile); a workgroup = 4 wavefronts = 2 channel halves x 2 pairs of 4-column strips; 75.8 KB of
// LDS: two workgroups per CU overlap each other's phases.  Per unit: 35 KiB of DMA, 66 + 66 transposing reads / writes, 336 fragment reads,
// 672 MFMAs, 128 accumulator writes, 32 tile reads.
#include "common.h"
#include "iseg_hip.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) void* dwm_lds_ptr;
typedef const __attribute__((address_space(1))) void* dwm_glb_ptr;

__device__ uint4 dwm_zero_page[4];      // 64 zero bytes (device globals are zero-initialised)

constexpr int TR = 16, TC = 16, CH = 32, KS = 7, WAVES = 8;
constexpr int RAW_ROWS = TR + 8, RAW_COLS = TC + 6, RAW_PITCH = 23;      // pixels; 24 rows = 6 chunks of 4 (rows 22, 23 meet zero taps only)
constexpr int RAW_PIXELS = RAW_ROWS * RAW_PITCH, RAW_PIECES = (RAW_PIXELS + 15) / 16;      // 35 one-KiB pieces
constexpr int NPW = (RAW_PIECES + WAVES - 1) / WAVES;      // DMA instructions per wavefront and tile: the SAME number for every wavefront (counted waits)
constexpr int RAW_BUF = NPW * WAVES * 1024;                // 40 KiB: pieces 35 .. 39 receive zeros nobody reads
constexpr int PLANE = 1248, COLP = 56;      // planar image: bytes per channel plane / per column (28 rows)
constexpr int P_BYTES = CH * PLANE;
constexpr int O_PITCH = TC * CH * 4 + 64;      // fp32 output tile row: 16 pixels x 128 B + 64 (the four row lanes of a block land 2-way = the 256-B minimum)
constexpr int O_BYTES = TR * O_PITCH;
constexpr int P_OFF = 2 * RAW_BUF, O_OFF = P_OFF + P_BYTES, LDS_BYTES = O_OFF + O_BYTES;
static_assert(RAW_COLS * COLP <= PLANE && LDS_BYTES <= 160 * 1024, "one workgroup of 8 wavefronts per CU");

typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

#define DWM_BARRIER()                                       \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                       \
        asm volatile("" ::: "memory");                      \
    } while (0)
// wait until at most N LDS operations are outstanding; the registers named are the ones the wait makes valid (keeps their consumers behind it)
#define DWM_GUARD7(N, A)                                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"((A)[0]), "+v"((A)[1]), "+v"((A)[2]), "+v"((A)[3]), "+v"((A)[4]), "+v"((A)[5]), "+v"((A)[6]) \
                 : "n"(N) : "memory")

// Every LDS access inside the unit loop is inline assembly: hipcc has no alias information between an LDS access it can see and the LDS-DMA
// (global_load_lds) in flight for the NEXT tile, and would put s_waitcnt vmcnt(0) in front of it -- the prefetch would never overlap anything
// (the same reason as in gemm_dma_tn.h / mlp_fused.hip).  The waits are written out; barriers are raw s_barrier (__syncthreads() carries a fence that
// drains vmcnt as well).
__global__ __launch_bounds__(64 * WAVES) void dwconv7_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const bf16_t* __restrict__ add,
                                                                bf16_t* __restrict__ y, int N, int H, int W, int C, int pad_t, int pad_l, int flip,
                                                                int tiles_h, int tiles_w, int slabs, int units_per_wg) {
    extern __shared__ __attribute__((aligned(1024))) char smem_dwm[];
    const unsigned lds0 = (unsigned)(uintptr_t)(dwm_lds_ptr)smem_dwm;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = N * tiles_h * tiles_w, units = ntiles * slabs;
    // Unit order (round 6).  A workgroup still walks a contiguous run of `units_per_wg` logical units, but (1) the runs are dealt XCD-contiguously
    // (workgroups b and b + 8 share an XCD under the round-robin placement: speed only), and (2) logical units are enumerated in BLOCKS of
    // units_per_wg tiles x all slabs, slab-major inside a block: the `slabs` workgroups of a block sit next to each other on one XCD, start together
    // and walk the SAME tiles in the same order, one channel slab each -- the 128-byte lines a 64-byte slab piece shares with its neighbour slab,
    // and the halo rows of vertically adjacent tiles, are then served by that XCD's L2 instead of being fetched again (PMC round 5: 245 MB per
    // launch at 128 x 128 x 96 for 100-150 MB algorithmic).  A run stays inside one slab, so the Toeplitz fragments are still built once.
    int lb = blockIdx.x;
    if (gridDim.x % 8 == 0) lb = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    const int u_begin = lb * units_per_wg, u_end = min(units, u_begin + units_per_wg);
    if (u_begin >= u_end) return;      // (workgroup-uniform)

    // ---- per-lane constants ----
    // DMA: piece p = wid + 8 i covers pixels 16 p .. 16 p + 15 of the raw tile, lane -> (pixel 16 p + lane / 4, 8-channel chunk lane % 4)
    int prc[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int pix = (wid + WAVES * i) * 16 + (lane >> 2);
        const int rr = pix / RAW_PITCH, cc = pix - rr * RAW_PITCH;
        prc[i] = (cc >= RAW_COLS || rr >= TR + KS - 1) ? -1 : (rr << 16) | cc;      // pad column, rows past the halo, pieces past the tile: zeros
    }
    const bf16_t* const zero = reinterpret_cast<const bf16_t*>(dwm_zero_page) + (lane & 3) * 8;
    // transposition: lane (g = lane / 16, q = (lane / 4) % 4, p = lane % 4) reads raw[4 R + q][2 cp + g / 2][16 (g & 1) + 4 p .. + 3]; afterwards lane
    // (g, i = lane % 16) holds rows 4 R .. 4 R + 3 of channel 16 (g & 1) + i at that column
    const int tg = lane >> 4, ti = lane & 15;
    const unsigned tr_src = lds0 + (((ti >> 2) * RAW_PITCH) + (tg >> 1)) * 64 + (tg & 1) * 32 + (ti & 3) * 8;
    const unsigned tr_dst = lds0 + P_OFF + ((tg & 1) * 16 + ti) * PLANE + (tg >> 1) * COLP;
    // compute: wavefront -> (channel half chg, 4-column strip), lane -> (channel chl, m / n = lane % 4)
    const int chg = wid & 1, strip = wid >> 1, chl = lane >> 2, mn = lane & 3;
    const unsigned frag_base = lds0 + P_OFF + (chg * 16 + chl) * PLANE + (mn + 4 * strip) * COLP;
    const unsigned out_base = lds0 + O_OFF + mn * O_PITCH + (4 * strip) * (CH * 4) + (chg * 16 + chl) * 4;

    // transposition tasks of this wavefront (row chunk R, column pair cp) = wid + 8 i: source / destination addresses, the same for every unit
    unsigned t_src[9], t_dst[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int task = wid + WAVES * i, R = task / 11, cp = task - R * 11;
        t_src[i] = tr_src + (4 * R * RAW_PITCH + 2 * cp) * 64;
        t_dst[i] = tr_dst + (2 * cp) * COLP + R * 8;
    }
    struct Unit {
        int slab, n, th, tw;
        int blk, k, tib;      // block of the enumeration, tile inside the block, tiles in this block (the last block may be short)
    };
    Unit cur;      // (workgroup-uniform: scalar registers)
    auto place = [&](Unit& q) {      // (n, th, tw) of tile blk * units_per_wg + k
        int t = q.blk * units_per_wg + q.k;
        q.tw = t % tiles_w;
        t /= tiles_w;
        q.th = t % tiles_h;
        q.n = t / tiles_h;
    };
    {
        const int bsz = units_per_wg * slabs;      // logical units per full block
        cur.blk = u_begin / bsz;
        const int r = u_begin - cur.blk * bsz;
        cur.tib = min(units_per_wg, ntiles - cur.blk * units_per_wg);
        cur.slab = r / cur.tib;
        cur.k = r - cur.slab * cur.tib;
        place(cur);
    }
    auto advance = [&](Unit& q) {
        if (++q.k == q.tib) {
            q.k = 0;
            if (++q.slab == slabs) {
                q.slab = 0;
                ++q.blk;
                q.tib = min(units_per_wg, ntiles - q.blk * units_per_wg);
            }
        }
        place(q);
    };
    int poff[NPW];      // element offset of this lane's pixel of piece i from the tile's first halo pixel: (rr W + cc) C, the same for every unit
#pragma unroll
    for (int i = 0; i < NPW; ++i) poff[i] = prc[i] < 0 ? 0 : ((prc[i] >> 16) * W + (prc[i] & 0xffff)) * C;
    // raw NHWC tile of unit q, HBM -> LDS buffer `buf`; !live: five zero-page pieces (the instruction count per wavefront never changes)
    auto issue_dma = [&](const Unit& q, int buf, bool live) {
        const int h0 = q.th * TR, w0 = q.tw * TC;
        const bf16_t* xb = x + ((int64_t)(q.n * H + h0 - pad_t) * W + (w0 - pad_l)) * C + q.slab * CH + (lane & 3) * 8;
        char* dst = smem_dwm + buf * RAW_BUF + wid * 1024;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            int rc = prc[i];
            asm volatile("" : "+v"(rc));
            const int rr = rc >> 16, cc = rc & 0xffff;
            const bool ok = live && rc >= 0 && (unsigned)(h0 - pad_t + rr) < (unsigned)H && (unsigned)(w0 - pad_l + cc) < (unsigned)W;
            const bf16_t* src = ok ? xb + poff[i] : zero;
#ifndef DWM_ABL_NODMA
            __builtin_amdgcn_global_load_lds((dwm_glb_ptr)src, (dwm_lds_ptr)(dst + i * WAVES * 1024), 16, 0, 0);
#else
            asm volatile("" ::"v"(src));
#endif
        }
    };

    s16x4 bf[KS][3];      // Toeplitz fragments of this lane's channel, rebuilt when the slab changes
    float bias_v = 0.f;
    int cur_slab = -1;
    int pending = 0;      // store instructions this wavefront issued behind the DMA that is in flight (wavefront-uniform)

    issue_dma(cur, 0, true);
    for (int u = u_begin; u < u_end; ++u) {
        const int buf = (u - u_begin) & 1;
        const int slab = cur.slab, n = cur.n, h0 = cur.th * TR, w0 = cur.tw * TC;
        const int c0 = slab * CH;
        advance(cur);      // cur = unit u + 1 from here on
        // tile u has landed: vmcnt retires in order, the only younger operations of this wavefront are the previous unit's stores
        if (pending >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (pending == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DWM_BARRIER();      // every wavefront's pieces are in; the previous unit's output tile has been read, its planar image consumed

        // the residual-branch gradient pieces of the output stage (bwd-data): requested in front of the next tile's DMA (older in vmcnt order)
        // (assembly loads, guarded by a counted wait in phase 4: a load hipcc can see gets s_waitcnt vmcnt(0) in front of its first use, i.e. behind the
        // next tile's DMA as well)
        u32x4_t addv[2] = {u32x4_t{0u, 0u, 0u, 0u}, u32x4_t{0u, 0u, 0u, 0u}};
        if (add) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int id = tid + 64 * WAVES * i, pix = id >> 2, chunk = id & 3;
                int oh = h0 + (pix >> 4), ow = w0 + (pix & 15);
                oh = oh < H ? oh : H - 1;
                ow = ow < W ? ow : W - 1;
                const bf16_t* ap = add + ((int64_t)(n * H + oh) * W + ow) * C + c0 + chunk * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(addv[i]) : "v"(ap) : "memory");
            }
        }
        issue_dma(cur, buf ^ 1, u + 1 < u_end);      // the next tile streams in behind everything below

        if (slab != cur_slab) {      // workgroup-uniform (its loads wait for the DMA just issued: once per slab)
            cur_slab = slab;
            const int ch = c0 + chg * 16 + chl;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    bf16x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ky = 4 * q + k - mn;
                        const int kyc = ky < 0 ? 0 : (ky > KS - 1 ? KS - 1 : ky);
                        const int tap = flip ? (KS - 1 - kyc) * KS + (KS - 1 - kx) : kyc * KS + kx;
                        const float wv = w[tap * C + ch];
                        v[k] = (bf16_t)((ky >= 0 && ky < KS) ? wv : 0.f);
                    }
                    bf[kx][q] = __builtin_bit_cast(s16x4, v);
                }
            bias_v = bias ? bias[ch] : 0.f;
        }

        // ---- phase 2: raw -> planar image (transposing reads); task = (row chunk R, column pair cp), 66 of them ----
        {
            const unsigned src0 = buf * RAW_BUF;
            u32x2_t tv[9] = {};
#pragma unroll
            for (int i = 0; i < 9; ++i) {
#ifndef DWM_ABL_NOTR
                if (wid + WAVES * i < 6 * 11) {      // (wavefront-uniform: EXEC stays full, which the transposing read needs)
                    const unsigned a = src0 + t_src[i];
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tv[i]) : "v"(a) : "memory");
                }
#endif
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]), "+v"(tv[6]), "+v"(tv[7]), "+v"(tv[8])
                         :: "memory");
#pragma unroll
            for (int i = 0; i < 9; ++i) {
#ifndef DWM_ABL_NOTR
                if (wid + WAVES * i < 6 * 11) asm volatile("ds_write_b64 %0, %1" ::"v"(t_dst[i]), "v"(tv[i]) : "memory");
#endif
            }
        }
        DWM_BARRIER();

        // ---- phase 3: banded products of one 4-column strip of 16 channels; accumulators -> fp32 output tile ----
        {
            u32x2_t a[6][KS];
#ifndef DWM_ABL_NOFRAG
#define DWM_RD(CK, KX) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[CK][KX]) : "v"(frag_base), "n"((KX) * COLP + (CK) * 8) : "memory")
#else
#define DWM_RD(CK, KX) a[CK][KX] = u32x2_t{(unsigned)lane, (unsigned)(CK * 7 + KX)}
#endif
#define DWM_RD7(CK) DWM_RD(CK, 0); DWM_RD(CK, 1); DWM_RD(CK, 2); DWM_RD(CK, 3); DWM_RD(CK, 4); DWM_RD(CK, 5); DWM_RD(CK, 6)
            // all 42 fragments of the strip (6 row chunks x 7 kernel columns), then the four row blocks as four INDEPENDENT accumulator chains issued
            // round-robin: a lone chain of 21 dependent 4x4x4 MFMAs ran at ~30 cycles per instruction instead of 8
            DWM_RD7(0); DWM_RD7(1); DWM_RD7(2); DWM_RD7(3); DWM_RD7(4); DWM_RD7(5);
            DWM_GUARD7(0, a[0]); DWM_GUARD7(0, a[1]); DWM_GUARD7(0, a[2]); DWM_GUARD7(0, a[3]); DWM_GUARD7(0, a[4]); DWM_GUARD7(0, a[5]);
#undef DWM_RD7
#undef DWM_RD
            f32x4_t d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#ifndef DWM_ABL_NOMFMA
                        d[j] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a[j + q][kx]), bf[kx][q], d[j], 0, 0, 0);
#else
                        asm volatile("" ::"v"(a[j + q][kx]));
#endif
                    }
            // (the bias add is a compiler-visible VALU instruction on purpose: hipcc inserts the MFMA -> VALU wait states there; an asm statement that
            // read the accumulators directly would be issued without them)
#pragma unroll
            for (int j = 0; j < 4; ++j) {      // row 4 j + n, columns 4 strip .. + 3
                const float o0 = d[j][0] + bias_v, o1 = d[j][1] + bias_v, o2 = d[j][2] + bias_v, o3 = d[j][3] + bias_v;
                const unsigned oa = out_base + (4 * j) * O_PITCH;
                asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:128\n\tds_write_b32 %0, %3 offset:256\n\tds_write_b32 %0, %4 offset:384"
                             ::"v"(oa), "v"(o0), "v"(o1), "v"(o2), "v"(o3) : "memory");
            }
        }
        DWM_BARRIER();

        // ---- phase 4: fp32 tile -> bf16 pieces (+ add) -> HBM; 1024 pieces, two per thread ----
        int issued = 0;
        f32x4_t ov[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 64 * WAVES * i, pix = id >> 2, chunk = id & 3;
            const unsigned oa = lds0 + O_OFF + (pix >> 4) * O_PITCH + (pix & 15) * (CH * 4) + chunk * 32;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=v"(ov[i][0]), "=v"(ov[i][1]) : "v"(oa) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ov[0][0]), "+v"(ov[0][1]), "+v"(ov[1][0]), "+v"(ov[1][1]) :: "memory");
        if (add) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(addv[0]), "+v"(addv[1]) : "n"(NPW) : "memory");      // younger: the next tile's NPW DMAs
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 64 * WAVES * i, pix = id >> 2, chunk = id & 3;
            const int orow = pix >> 4, ocol = pix & 15;
            float v[8] = {ov[i][0][0], ov[i][0][1], ov[i][0][2], ov[i][0][3], ov[i][1][0], ov[i][1][1], ov[i][1][2], ov[i][1][3]};
            if (add) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, addv[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)av[e];
            }
            const int oh = h0 + orow, ow = w0 + ocol;
            if (oh < H) ++issued;      // (a wavefront's 16 pixels are one tile row: wavefront-uniform; w0 < W, so some column is inside)
#ifdef DWM_ABL_NOOUT
            if (oh < H && ow < W && v[0] == 1234.5f)
#else
            if (oh < H && ow < W)
#endif
                store8<bf16_t>(y + ((int64_t)(n * H + oh) * W + ow) * C + c0 + chunk * 8, v);
        }
        pending = __builtin_amdgcn_readfirstlane(issued);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dummy DMA of the last iteration must not outlive the workgroup's LDS
}

int dw_mfma_mode() {
    static const int v = [] {
        const char* e = getenv("ISEG_DW_MFMA");
        return e ? atoi(e) : 1;
    }();
    return v;
}

}  // namespace

static bool dwm_eligible(const void* x, const void* add, const void* y, int C, int K, int dil) {
    return K == 7 && dil == 1 && C % 32 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)add) & 15) == 0;
}

static bool dwm_launch(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int pad_t, int pad_l,
                       int flip, hipStream_t s) {
    const int tiles_h = (H + TR - 1) / TR, tiles_w = (W + TC - 1) / TC, slabs = C / CH;
    const int64_t units = (int64_t)N * tiles_h * tiles_w * slabs;
    if (units >= (1ll << 30)) return false;
    static const bool raised = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv7_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
    }();
    if (!raised) return false;
    // one resident workgroup per CU (155 KB of LDS: two raw tiles, planar image, output tile); contiguous runs of units per workgroup (a run mostly
    // stays inside one channel slab: the Toeplitz fragments are rebuilt only when the slab changes)
    const int64_t slots = 256;
    const int64_t per = (units + slots - 1) / slots;
    const int64_t nwg = (units + per - 1) / per;
    hipLaunchKernelGGL(dwconv7_mfma_kernel, dim3((unsigned)nwg), dim3(64 * WAVES), LDS_BYTES, s, (const bf16_t*)x, w, bias, (const bf16_t*)add, (bf16_t*)y, N, H,
                       W, C, pad_t, pad_l, flip, tiles_h, tiles_w, slabs, (int)per);
    return true;
}

// Automatic route of iseg_dwconv2d_fwd (dwconv.hip falls through to the VALU kernels when this returns false).  Measured, 16 images, us, VALU DMA
// kernel -> this one (tools/kbench_dw_mfma.py, profiles/r05_dw_mfma.txt): 128 x 128 x 96 forward 54.0 -> 48.9, data gradient 61.6 -> 53.4;
// 64 x 64 x 192 31.2 -> 32.5; 32 x 32 x 384 17.3 -> 21.3; 16 x 16 x 768 11.2 -> 13.5.  v_mfma_f32_4x4x4_16b_bf16 issues every 16 cycles (64 MAC per
// clock and SIMD, 37 of them taps): 1.4x the packed-FMA kernels' arithmetic rate, and the kernel's phases (fill, transposition, products, write-out)
// do not overlap inside its one workgroup per CU -- it wins where a workgroup walks >= 8 tiles.  ISEG_DW_MFMA: 0 = never, 1 = automatic (default),
// 2 = whenever eligible.
bool iseg_dwconv7_mfma_launch(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int K, int dil,
                              int pad_t, int pad_l, int flip, hipStream_t s) {
    const int mode = dw_mfma_mode();
    if (!mode || !dwm_eligible(x, add, y, C, K, dil)) return false;
    const int64_t units = (int64_t)N * ((H + TR - 1) / TR) * ((W + TC - 1) / TC) * (C / CH);
    if (mode == 1 && units < 2048) return false;
    return dwm_launch(x, w, bias, add, y, N, H, W, C, pad_t, pad_l, flip, s);
}

extern "C" int iseg_dwconv2d7_mfma(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int pad_t,
                                   int pad_l, int flip, hipStream_t stream) {
    ISEG_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0, "iseg_dwconv2d7_mfma: bad arguments");
    ISEG_REQUIRE((int64_t)N * H * W * C < (1ll << 31), "iseg_dwconv2d7_mfma: more than 2^31 elements");
    if (!dwm_eligible(x, add, y, C, 7, 1) || !dwm_launch(x, w, bias, add, y, N, H, W, C, pad_t, pad_l, flip, stream)) {
        iseg_set_error("iseg_dwconv2d7_mfma: needs bf16 storage, C %% 32 == 0 (C = %d) and 16-byte aligned tensors", C);
        return ISEG_ERR_UNSUPPORTED;
    }
    return iseg_check_launch("iseg_dwconv2d7_mfma");
}
