// bf16 MFMA GEMM kernel template + epilogue shared by gemm.hip (C ABI, fp32 path, split-K reducer) and the three
// orientation translation units gemm_nn.hip / gemm_nt.hip / gemm_tn.hip (split so hipcc compiles them in parallel).
//
// Structure (per 256-thread workgroup = 4 waves, one 128 x BN output tile):
//   * operand tiles are staged HBM -> registers (16-B lanes, coalesced along each operand's contiguous dimension)
//     -> LDS in their native orientation; the next K-tile's global loads are issued before the current tile's MFMAs so
//     they are in flight during compute (single LDS buffer: LDS stays <= 53 KB so 3 workgroups fit a CU);
//   * K-contiguous operands (activations [M][K], weights [N][K]) become fragments with one ds_read_b128 per lane,
//     MN-contiguous operands (Keras [K][N] kernels, transposed activations for wgrad) with two ds_read_b64_tr_b16;
//   * BK = 64 (two MFMA k-steps per barrier pair); problems with K <= 96 are done in ONE tile (BK = 64 or 96), so the
//     HBM-bound ConvNeXt C=96 layers issue all their loads up front and hit a single barrier;
//   * accumulators -> per-wave LDS slab (fp32) -> 8-column rows -> fused epilogue -> 16-B coalesced stores.
#pragma once
#include "common.h"
#include "iseg_hip.h"

namespace iseg_mm {

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
    float* colsum_out;      // wgrad only: column sums of B over the reduction (bias gradient) through a virtual ones-row of A
    int colsum_accumulate;
    // strided batch: problem z = blockIdx.z
    int batch_inner;
    int64_t sa_outer, sa_inner, sb_outer, sb_inner, sd_outer, sd_inner;
    int pre_deriv;          // pre_out receives gelu'(pre) instead of pre (act == GELU)
    int64_t b_group_rows, b_group_stride;      // B per row group (LDS-DMA kernel only): rows [i*b_group_rows, ...) use B + i*b_group_stride
    int bias_rowscaled;     // the row factor multiplies the bias, not the result (iseg_gemm_args.bias_rowscaled)
    __device__ __forceinline__ int64_t off_a(int z) const { return (z / batch_inner) * sa_outer + (z % batch_inner) * sa_inner; }
    __device__ __forceinline__ int64_t off_b(int z) const { return (z / batch_inner) * sb_outer + (z % batch_inner) * sb_inner; }
    __device__ __forceinline__ int64_t off_d(int z) const { return (z / batch_inner) * sd_outer + (z % batch_inner) * sd_inner; }
};

template <class TO>
__device__ __forceinline__ float epi_apply(const Epi& e, float acc, int64_t m, int64_t n, const TO* D, int64_t ldd) {
    float v = acc * e.alpha;
    const bool brs = e.bias_rowscaled && e.rowscale;
    if (e.bias) v += brs ? e.bias[n] * e.rowscale[m / e.rows_per_group] : e.bias[n];
    if (e.pre_out) reinterpret_cast<TO*>(e.pre_out)[m * e.ldp + n] = from_f32<TO>(e.pre_deriv ? gelu_erf_grad(v) : v);
    if (e.act == ISEG_ACT_RELU) v = fmaxf(v, 0.f);
    else if (e.act == ISEG_ACT_GELU) v = gelu_erf(v);
    else if (e.act == ISEG_ACT_GELU_GRAD) v *= gelu_erf_grad(to_f32(reinterpret_cast<const TO*>(e.aux)[m * e.ldaux + n]));
    else if (e.act == ISEG_ACT_RELU_GRAD) v = to_f32(reinterpret_cast<const TO*>(e.aux)[m * e.ldaux + n]) > 0.f ? v : 0.f;
    else if (e.act == ISEG_ACT_MUL_AUX) v *= to_f32(reinterpret_cast<const TO*>(e.aux)[m * e.ldaux + n]);
    if (e.colscale) v *= e.colscale[n];
    if (e.rowscale && !brs) v *= e.rowscale[m / e.rows_per_group];
    if (e.residual) v += to_f32(reinterpret_cast<const TO*>(e.residual)[m * e.ldr + n]);
    if (e.accumulate) v += to_f32(D[m * ldd + n]);
    return v;
}

// 8 consecutive columns of one row, all pointers 16-B friendly (checked by the caller)
template <class TO>
__device__ __forceinline__ void epi_apply8(const Epi& e, float* v, int64_t m, int64_t n, TO* D, int64_t ldd) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
    const bool brs = e.bias_rowscaled && e.rowscale;      // (the row factor goes on the bias: see iseg_gemm_args.bias_rowscaled)
    if (e.bias) {
        float b[8];
        load8<float>(e.bias + n, b);
        const float bs = brs ? e.rowscale[m / e.rows_per_group] : 1.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(b[i], bs, v[i]);
    }
    constexpr bool FAST = sizeof(TO) == 2;  // bf16 storage: approximation error << output rounding; fp32 parity path: libm erf
    if (e.pre_out && !e.pre_deriv) store8<TO>(reinterpret_cast<TO*>(e.pre_out) + m * e.ldp + n, v);
    if (e.act == ISEG_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (e.act == ISEG_ACT_GELU) {
        if (e.pre_out && e.pre_deriv) {
            float d[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (FAST) gelu_sig_both(v[i], v[i], d[i]);
                else {
                    d[i] = gelu_erf_grad(v[i]);
                    v[i] = gelu_erf(v[i]);
                }
            }
            store8<TO>(reinterpret_cast<TO*>(e.pre_out) + m * e.ldp + n, d);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = FAST ? gelu_poly(v[i]) : gelu_erf(v[i]);
        }
    } else if (e.act == ISEG_ACT_MUL_AUX) {
        float a[8];
        load8<TO>(reinterpret_cast<const TO*>(e.aux) + m * e.ldaux + n, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= a[i];
    } else if (e.act == ISEG_ACT_GELU_GRAD) {
        float a[8];
        load8<TO>(reinterpret_cast<const TO*>(e.aux) + m * e.ldaux + n, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= FAST ? gelu_poly_grad(a[i]) : gelu_erf_grad(a[i]);
    } else if (e.act == ISEG_ACT_RELU_GRAD) {
        float a[8];
        load8<TO>(reinterpret_cast<const TO*>(e.aux) + m * e.ldaux + n, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = a[i] > 0.f ? v[i] : 0.f;
    }
    if (e.colscale) {
        float c[8];
        load8<float>(e.colscale + n, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= c[i];
    }
    if (e.rowscale && !brs) {
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

// raw 8-element vectors (kept packed in registers while several epilogue rows are prefetched)
template <class T> struct Raw8;
template <> struct Raw8<bf16_t> {
    bf16x8 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
};
template <> struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ float get(int i) const {
        return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : i == 3 ? a.w : i == 4 ? b.x : i == 5 ? b.y : i == 6 ? b.z : b.w;
    }
};

template <class TO> struct EpiPrefetch {
    Raw8<TO> aux, res, old;
    __device__ __forceinline__ void load(const Epi& e, int64_t m, int64_t n, const TO* D, int64_t ldd) {
        if (e.act == ISEG_ACT_GELU_GRAD || e.act == ISEG_ACT_RELU_GRAD || e.act == ISEG_ACT_MUL_AUX)
            aux.load(reinterpret_cast<const TO*>(e.aux) + m * e.ldaux + n);
        if (e.residual) res.load(reinterpret_cast<const TO*>(e.residual) + m * e.ldr + n);
        if (e.accumulate) old.load(D + m * ldd + n);
    }
};

// epi_apply8 with the global operands already in registers
template <class TO>
__device__ __forceinline__ void epi_finish8(const Epi& e, float* v, const EpiPrefetch<TO>& pf, int64_t m, int64_t n, TO* D, int64_t ldd) {
    constexpr bool FAST = sizeof(TO) == 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
    const bool brs = e.bias_rowscaled && e.rowscale;      // (the row factor goes on the bias: see iseg_gemm_args.bias_rowscaled)
    if (e.bias) {
        float b[8];
        load8<float>(e.bias + n, b);
        const float bs = brs ? e.rowscale[m / e.rows_per_group] : 1.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(b[i], bs, v[i]);
    }
    if (e.pre_out && !e.pre_deriv) store8<TO>(reinterpret_cast<TO*>(e.pre_out) + m * e.ldp + n, v);
    if (e.act == ISEG_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (e.act == ISEG_ACT_GELU) {
        if (e.pre_out && e.pre_deriv) {      // Phi(v) and exp(-v^2/2) serve both gelu(v) and gelu'(v)
            float d[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (FAST) gelu_sig_both(v[i], v[i], d[i]);
                else {
                    d[i] = gelu_erf_grad(v[i]);
                    v[i] = gelu_erf(v[i]);
                }
            }
            store8<TO>(reinterpret_cast<TO*>(e.pre_out) + m * e.ldp + n, d);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = FAST ? gelu_poly(v[i]) : gelu_erf(v[i]);
        }
    } else if (e.act == ISEG_ACT_MUL_AUX) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= pf.aux.get(i);
    } else if (e.act == ISEG_ACT_GELU_GRAD) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= FAST ? gelu_poly_grad(pf.aux.get(i)) : gelu_erf_grad(pf.aux.get(i));
    } else if (e.act == ISEG_ACT_RELU_GRAD) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = pf.aux.get(i) > 0.f ? v[i] : 0.f;
    }
    if (e.colscale) {
        float c[8];
        load8<float>(e.colscale + n, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= c[i];
    }
    if (e.rowscale && !brs) {
        const float s = e.rowscale[m / e.rows_per_group];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= s;
    }
    if (e.residual) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += pf.res.get(i);
    }
    if (e.accumulate) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += pf.old.get(i);
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

// (output tile, K split) of a workgroup.  The remap runs over the FLATTENED grid (split-major): the workgroups that share an XCD get a
// contiguous run of (split, tile) pairs, i.e. all tiles of the same K-slice -- which read the same rows of both operands -- sit on one
// L2.  Remapping only inside a split (round 1) dealt the tiles of one K-slice over all eight XCDs: the stage-3 weight gradient
// (36 tiles x 13 splits) moved 212 MB for 63 MB of operands (profiles/r02_pmc.json).
__device__ __forceinline__ void tile_and_split(int ntiles, int& tile, int& split) {
    if (gridDim.y > 1) {
        const int v = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, ntiles * gridDim.y);
        tile = v % ntiles;
        split = v / ntiles;
    } else {
        tile = xcd_remap(blockIdx.x, ntiles);
        split = 0;
    }
}

constexpr int KPAD = 8;   // K-contiguous tiles: row stride BK + 8 elements (conflict-free ds_read_b128 for BK 32/64/96/128)
constexpr int MNPAD = 8;  // MN-contiguous tiles: row stride B{M,N} + 8 elements

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

template <int ROWS, bool KC, int BK> struct TileGeom {
    static constexpr int stride = KC ? (BK + KPAD) : (ROWS + MNPAD);
    static constexpr int elems = KC ? ROWS * stride : BK * stride;
    static constexpr int chunks = ROWS * BK / 8;  // 16-B chunks in the tile
};

// Load one 16-B chunk (8 bf16) of an operand tile from global memory, zero-filled out of range.
//  KC:  element (r, k) at base[r * ld + k]   (r along M or N)
//  !KC: element (r, k) at base[k * ld + r]
template <bool KC>
__device__ __forceinline__ bf16x8 load_chunk(const bf16_t* __restrict__ base, int64_t ld, int64_t r0, int64_t k0, int64_t R,
                                             int64_t Kend, bool vec) {
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

template <int ROWS, bool KC, int NTHREADS, int BK> struct Stager {
    using G = TileGeom<ROWS, KC, BK>;
    static constexpr int PER_THREAD = (G::chunks + NTHREADS - 1) / NTHREADS;
    bf16x8 regs[PER_THREAD];

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
                                         int64_t Kend, bool vec, int tid, bool ones_row = false) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NTHREADS;
            if (G::chunks % NTHREADS == 0 || c < G::chunks) {
                int r, k;
                decode(c, r, k);
                regs[i] = load_chunk<KC>(base, ld, row0 + r, k0 + k, R, Kend, vec);
                // virtual row R of an MN-contiguous A filled with ones: output row R becomes sum_k B(k, :)
                if (!KC && ones_row && row0 + r == R && k0 + k < Kend) regs[i][0] = (bf16_t)1.0f;
            }
        }
    }
    // operand transform on the staged registers: A := gelu(A).  Lets the pre-activation be the only [M,4C] tensor that
    // ConvNeXt's MLP ever writes (backbones/convnext.py:51-54): pwconv2 and the pwconv2 weight gradient re-derive gelu(h).
    __device__ __forceinline__ void apply_gelu() {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) regs[i][u] = (bf16_t)gelu_poly((float)regs[i][u]);
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

// 16(row) x 32(k) MFMA operand fragment of k-step `ks` from an LDS tile; `r0` = first tile row of the fragment.
template <int ROWS, bool KC, int BK>
__device__ __forceinline__ bf16x8 read_frag(const bf16_t* lds, int r0, int ks, int lane) {
    using G = TileGeom<ROWS, KC, BK>;
    if (KC) {
        // lane holds row (lane&15), k = 32*ks + 8*(lane>>4) .. +7 : one ds_read_b128
        return *reinterpret_cast<const bf16x8*>(lds + (r0 + (lane & 15)) * G::stride + 32 * ks + 8 * (lane >> 4));
    } else {
        // tile is [k][row]; ds_read_b64_tr_b16: lane 4q+p of each 16-lane group addresses LDS row q,
        // columns 4p..4p+3, and receives column (lane&15) of the four rows.
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const bf16_t* a0 = lds + (32 * ks + 8 * g + q) * G::stride + r0 + 4 * p;
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

// the row of D (and of the fused operands) that GEMM row m lands in: the identity everywhere except the phase form of the strided
// convolution data gradient (conv_igemm.hip), whose rows are the pixels of one stride phase scattered back into the NHWC map
struct IdentityRows {
    __device__ __forceinline__ int64_t operator()(int64_t m) const { return m; }
};

// ---- epilogue: accumulators -> per-wave LDS slab (fp32) -> 8-column coalesced rows ----
// Shared by both main loops.  `smem` must be free (every wave past its last fragment read) and hold WM*WN*32*(FN*16+4) floats.
template <int WM, int WN, int FM, int FN, class TO, class RowMap = IdentityRows>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[FM][FN], char* smem, TO* __restrict__ D, int64_t ldd, int64_t M, int64_t N,
                                              int64_t m0, int64_t n0, int wm, int wn, int wid, int lane, float* __restrict__ slabs,
                                              const Epi& epi, bool ones_row, int vecD, int ksplit, const RowMap rowmap = RowMap()) {
    constexpr int TM = FM * 16, TN = FN * 16;
    constexpr int EPI_ROWS = 32;
    constexpr int EPI_STRIDE = TN + 4;
    float* const ew = reinterpret_cast<float*>(smem) + wid * EPI_ROWS * EPI_STRIDE;
    const bool split = slabs != nullptr;
    const int64_t slab_rows = M + (ones_row ? 1 : 0);
    float* const slab = split ? slabs + (int64_t)ksplit * slab_rows * N : nullptr;
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
                for (int j = 0; j < FN; ++j)      // transposed accumulators: row = lane & 15, four consecutive columns per lane
                    *reinterpret_cast<float4*>(ew + (fi * 16 + (lane & 15)) * EPI_STRIDE + j * 16 + (lane >> 4) * 4) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        __builtin_amdgcn_wave_barrier();  // the slab is wave-private and LDS ops of a wave complete in order
        // all of this pass's rows: LDS reads and global operand loads first (kept packed), then the arithmetic and the stores
        constexpr int ITERS = (EPI_ROWS * CPR + 63) / 64;
        constexpr int PF = 1;  // rows whose operands are in flight together (2 was measured: no gain, costs occupancy)
#pragma unroll
        for (int g0 = 0; g0 < ITERS; g0 += PF) {
            float v[PF][8];
            EpiPrefetch<TO> pf[PF];
            int64_t mm[PF], nn[PF];
            int kind[PF];  // 0 skip, 1 split slab, 2 vector epilogue, 3 scalar epilogue
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int c = lane + (g0 + q) * 64;
                const int rr = c / CPR, cc = (c % CPR) * 8;
                mm[q] = m0 + wm * TM + ps * EPI_ROWS + rr;
                nn[q] = n0 + wn * TN + cc;
                kind[q] = 0;
                if (g0 + q < ITERS && c < EPI_ROWS * CPR && ps * EPI_ROWS + rr < TM && mm[q] < slab_rows && nn[q] < N) {
                    const float* src = ew + rr * EPI_STRIDE + cc;
                    *reinterpret_cast<float4*>(v[q]) = *reinterpret_cast<const float4*>(src);
                    *reinterpret_cast<float4*>(v[q] + 4) = *reinterpret_cast<const float4*>(src + 4);
                    kind[q] = split ? 1 : (mm[q] == M ? 4 : ((vecD && nn[q] + 8 <= N) ? 2 : 3));
                    if (kind[q] == 2 || kind[q] == 3) mm[q] = rowmap(mm[q]);
                    if (kind[q] == 2) pf[q].load(epi, mm[q], nn[q], D, ldd);
                }
            }
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int64_t m = mm[q], n = nn[q];
                if (kind[q] == 1) {
                    float* dst = slab + m * N + n;
                    if (n + 8 <= N && (N % 4 == 0)) {
                        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(v[q]);
                        *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(v[q] + 4);
                    } else {
                        for (int u = 0; u < 8 && n + u < N; ++u) dst[u] = v[q][u];
                    }
                } else if (kind[q] == 2) {
                    epi_finish8<TO>(epi, v[q], pf[q], m, n, D, ldd);
                } else if (kind[q] == 3) {
                    for (int u = 0; u < 8 && n + u < N; ++u)
                        D[m * ldd + n + u] = from_f32<TO>(epi_apply<TO>(epi, v[q][u], m, n + u, D, ldd));
                } else if (kind[q] == 4) {  // the ones-row: column sums of B
                    for (int u = 0; u < 8 && n + u < N; ++u)
                        epi.colsum_out[n + u] = v[q][u] + (epi.colsum_accumulate ? epi.colsum_out[n + u] : 0.f);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int WM, int WN, int FM, int FN, bool AKC, bool BKC, int BK, class TO>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                 const bf16_t* __restrict__ B, int64_t ldb, TO* __restrict__ D,
                                                                 int64_t ldd, int64_t M, int64_t N, int64_t K, int tiles_n,
                                                                 int ntiles, int64_t k_per_split, float* __restrict__ slabs,
                                                                 Epi epi, int vecA, int vecB, int vecD, int a_act) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
    constexpr int TM = FM * 16, TN = FN * 16;
    constexpr int KS = BK / 32;
    using GA = TileGeom<BM, AKC, BK>;
    using GB = TileGeom<BN, BKC, BK>;
    constexpr int STAGE_ELEMS = GA::elems + GB::elems;
    constexpr int EPI_ROWS = 32;  // rows of a wave tile staged per epilogue pass
    constexpr int EPI_STRIDE = TN + 4;
    constexpr int EPI_BYTES = WM * WN * EPI_ROWS * EPI_STRIDE * 4;
    constexpr int LDS_BYTES = (STAGE_ELEMS * 2 > EPI_BYTES) ? STAGE_ELEMS * 2 : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    bf16_t* const lds = reinterpret_cast<bf16_t*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    if (gridDim.z > 1) {   // strided batch
        A += epi.off_a(blockIdx.z);
        B += epi.off_b(blockIdx.z);
        D += epi.off_d(blockIdx.z);
    }

    int t, ksplit;
    tile_and_split(ntiles, t, ksplit);
    const int tile_n = t % tiles_n, tile_m = t / tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int64_t kbeg = (int64_t)ksplit * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    const int nk = (int)((kend - kbeg + BK - 1) / BK);

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Stager<BM, AKC, NT, BK> sa;
    Stager<BN, BKC, NT, BK> sb;

    const bool ones_row = !AKC && epi.colsum_out != nullptr;
    sa.load(A, lda, m0, kbeg, M, kend, vecA != 0, tid, ones_row);
    sb.load(B, ldb, n0, kbeg, N, kend, vecB != 0, tid);
    if (a_act == ISEG_ACT_GELU) sa.apply_gelu();
    sa.store(lds, tid);
    sb.store(lds + GA::elems, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {  // next tile's HBM loads fly during this tile's MFMAs
            const int64_t k0 = kbeg + (int64_t)(kt + 1) * BK;
            sa.load(A, lda, m0, k0, M, kend, vecA != 0, tid, ones_row);
            sb.load(B, ldb, n0, k0, N, kend, vecB != 0, tid);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = read_frag<BM, AKC, BK>(lds, wm * TM + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = read_frag<BN, BKC, BK>(lds + GA::elems, wn * TN + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    // B fragment first: the accumulator holds the transposed product, i.e. lane (g, c) owns output row 16*i + c and
                    // columns 16*j + 4*g .. + 3 -- the epilogue slab is then filled with 16-byte rows instead of scattered dwords
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // every wave is done reading the tile (also fences the epilogue's reuse of the LDS)
        if (more) {
            if (a_act == ISEG_ACT_GELU) sa.apply_gelu();
            sa.store(lds, tid);
            sb.store(lds + GA::elems, tid);
            __syncthreads();
        }
    }

    tile_epilogue<WM, WN, FM, FN, TO>(acc, smem, D, ldd, M, N, m0, n0, wm, wn, wid, lane, slabs, epi, ones_row, vecD, ksplit);
}

// experiment knob (read once)
int long_k_tile();  // ISEG_GEMM_BK in {64,128}: K-tile depth for reductions >= 256 (default 128)
int tile_waves();   // ISEG_GEMM_WAVES in {4,8,16}: workgroup size of the 128x128 tile (default 8)

template <int WM, int WN, int FM, int FN, bool AKC, bool BKC, int BK, class TO>
void launch_bf16(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t k_per_split, float* slabs, hipStream_t s) {
    constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
    // (the virtual ones-row of the weight-gradient orientation is output row M: one more tile row when M is a multiple of the tile)
    const int tiles_m = (int)ceil_div64(g->M + ((!AKC && epi.colsum_out) ? 1 : 0), BM), tiles_n = (int)ceil_div64(g->N, BN);
    const int ntiles = tiles_m * tiles_n;
    const bf16_t* A = (const bf16_t*)g->A;
    const bf16_t* B = (const bf16_t*)g->B;
    const int vecA = ((uintptr_t)A % 16 == 0) && (g->lda % 8 == 0);
    const int vecB = ((uintptr_t)B % 16 == 0) && (g->ldb % 8 == 0);
    int vecD = ((uintptr_t)g->D % 16 == 0) && (g->ldd % 8 == 0);
    if (g->residual) vecD = vecD && ((uintptr_t)g->residual % 16 == 0) && (g->ldr % 8 == 0);
    if (g->aux) vecD = vecD && ((uintptr_t)g->aux % 16 == 0) && (g->ldaux % 8 == 0);
    if (g->pre_out) vecD = vecD && ((uintptr_t)g->pre_out % 16 == 0) && (g->ldp % 8 == 0);
    if (g->bias) vecD = vecD && ((uintptr_t)g->bias % 16 == 0);
    if (g->colscale) vecD = vecD && ((uintptr_t)g->colscale % 16 == 0);
    const int batch = g->batch > 1 ? g->batch : 1;
    int vA = vecA, vB = vecB, vD = vecD;
    if (batch > 1) {   // every problem of the batch must keep the 16-byte alignment the vector paths assume
        vA = vA && g->sa_outer % 8 == 0 && g->sa_inner % 8 == 0;
        vB = vB && g->sb_outer % 8 == 0 && g->sb_inner % 8 == 0;
        vD = vD && g->sd_outer % 8 == 0 && g->sd_inner % 8 == 0;
    }
    dim3 grid(ntiles, nsplit, batch);
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, FM, FN, AKC, BKC, BK, TO>), grid, dim3(WM * WN * 64), 0, s, A, g->lda, B, g->ldb,
                       (TO*)g->D, g->ldd, g->M, g->N, g->K, tiles_n, ntiles, k_per_split, slabs, epi, vA, vB, vD, g->a_act);
}

template <bool AKC, bool BKC, int BK, class TO>
void dispatch_tile(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    const int64_t N = g->N;
    // all tiles are 128 rows; 8 waves per workgroup measured ~2x faster than 4 on these epilogue-heavy, HBM-bound shapes
    if (N <= 32) launch_bf16<8, 1, 1, 2, AKC, BKC, BK, TO>(g, epi, nsplit, kps, slabs, s);                               // 128x32
    else if (N <= 64) launch_bf16<4, 2, 2, 2, AKC, BKC, BK, TO>(g, epi, nsplit, kps, slabs, s);                          // 128x64
    else if (N % 128 != 0 && N % 96 == 0) launch_bf16<4, 2, 2, 3, AKC, BKC, BK, TO>(g, epi, nsplit, kps, slabs, s);      // 128x96
    else if (tile_waves() == 4) launch_bf16<2, 2, 4, 4, AKC, BKC, BK, TO>(g, epi, nsplit, kps, slabs, s);                // 128x128
    else if (tile_waves() == 16) launch_bf16<4, 4, 2, 2, AKC, BKC, BK, TO>(g, epi, nsplit, kps, slabs, s);
    else launch_bf16<2, 4, 4, 2, AKC, BKC, BK, TO>(g, epi, nsplit, kps, slabs, s);
}

// one K-tile for short reductions (no loop, all loads issued up front), BK = 64 otherwise
template <bool AKC, bool BKC, class TO>
void dispatch_bk(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    const int64_t kspan = kps < g->K ? kps : g->K;
    if (AKC && kspan > 64 && kspan <= 96) dispatch_tile<AKC, BKC, 96, TO>(g, epi, nsplit, kps, slabs, s);
    else if (kspan >= 256 && long_k_tile() == 128) dispatch_tile<AKC, BKC, 128, TO>(g, epi, nsplit, kps, slabs, s);  // long reductions:
    else dispatch_tile<AKC, BKC, 64, TO>(g, epi, nsplit, kps, slabs, s);   // twice the bytes in flight per barrier pair
}

// implemented in gemm_nn.hip (A K-contig, B N-contig), gemm_nt.hip (both K-contig), gemm_tn.hip (both MN-contig)
void gemm_bf16_nn(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s);
void gemm_bf16_nt(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s);
void gemm_bf16_tn(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s);
int gemm_bf16_tn_pair_split(const iseg_gemm_args* g0, const iseg_gemm_args* g1);
void gemm_bf16_tn_pair(const iseg_gemm_args* g0, float* slabs0, const iseg_gemm_args* g1, float* slabs1, int nsplit, int64_t kps, hipStream_t s);

}  // namespace iseg_mm
