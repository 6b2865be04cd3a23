// Fused ConvNeXt MLP (backbones/convnext.py:51-63 of the reference):
//     out = x + rowscale[sample] * gamma * (gelu(y2 @ W1 + b1) @ W2 + b2)
// for the wide, shallow stages (C = 96 / 192; M = 262144 / 65536 rows at the flagship size), where the [M, 4C] hidden tensor is
// 4x the activation and the un-fused pair of GEMMs is a pure HBM stream (two writes and two reads of 200 MB per block at stage 0).
// Here the hidden tile never leaves the CU.
//
// Everything is computed TRANSPOSED so that the first product's accumulator is directly the second product's MFMA operand
// (v_mfma_f32_32x32x16_bf16: the accumulator has its column on the lane and its rows in the 16 registers; a following MFMA that
// sums over those rows takes registers 8s..8s+7 as the B fragment of k-step s, no lane movement, no LDS):
//     H^T[hid][m] = W1^T[hid][c] . y2^T[c][m]        A = W1 slab (transposed LDS read), B = y2 rows (registers, loaded once)
//     G^T        = gelu(H^T + b1)                     in registers, rounded to bf16
//     O^T[c][m] += W2^T[c][hid] . G^T[hid][m]         A = W2 slab (transposed LDS read), B = G^T
// A wavefront owns 32 rows (m) and ALL C output channels (C/32 accumulator blocks); a workgroup = 8 wavefronts = 256 rows.  The
// weights stream through a 3-stage LDS ring in slabs of 32 hidden units (global_load_lds_dwordx4, counted vmcnt, one raw s_barrier
// per stage -- the ring protocol of gemm_dma.h).  Both weight images are stored [k][32 MN] with 64-byte rows: a transposed read
// (ds_read_b64_tr_b16) of one 32-lane half covers 4 rows x 64 B = all 64 banks once, so no swizzle is needed:
//     W1 slab  [c = 0..C-1][32 hid]                    straight 64-byte pieces of the Keras [C][4C] kernel rows
//     W2 slab  [cb = 0..C/32-1][32 hid][32 c]          64-byte pieces of the Keras [4C][C] kernel rows, regrouped by the DMA
// Per slab and wavefront: C/16 + C/16 MFMAs (32 cycles each) and 16 gelu evaluations per lane (gelu_sig: ~44 issue cycles each)
// -> VALU-bound at C = 96, balanced at C = 192; two wavefronts per SIMD let one's gelu overlap the other's MFMAs.
// The backward kernels (below) recompute H^T from y2 instead of reading a saved [M, 4C] tensor.
#include "common.h"
#include "iseg_hip.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

// A fragment (32 rows x 16 k) of v_mfma_f32_32x32x16_bf16 from a [k][32 MN] image with 64-byte rows: lane (r = l & 31, h = l >> 5)
// needs k = 8h + j (j = 0..7) of MN row r.  One ds_read_b64_tr_b16 hands each lane 4 consecutive k rows of its column; `a0` is the
// lane's address for the first four, `second` the byte distance to the next four.
__device__ __forceinline__ bf16x8 tr_frag(const char* a0, int second) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + second));
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = lo[i];
        f[4 + i] = hi[i];
    }
    return f;
}

constexpr int MLP_NS = 3;                 // ring stages
constexpr int OUT_SLAB = 32 * 36 * 4;     // per-wavefront epilogue slab: 32 rows x (32 + 4 pad) floats

template <int C, int SUB> struct MlpGeom {
    static constexpr int HID = 4 * C, KK = C / 16, CB = C / 32;
    static constexpr int SLAB = 2 * C * 64;            // bytes per 32 hidden units: W1 image (C*64), then W2 image (C*64)
    static constexpr int STAGE = SUB * SLAB;
    static constexpr int PIECES = STAGE / 1024, PPW = PIECES / 8;
    static constexpr int NST = HID / (32 * SUB);
    static constexpr int RING = MLP_NS * STAGE > 8 * OUT_SLAB ? MLP_NS * STAGE : 8 * OUT_SLAB;
    static constexpr int LDS = RING + HID * 4;         // + b1 as floats
    static_assert(C % 32 == 0 && PIECES % 8 == 0 && HID % (32 * SUB) == 0, "geometry");
};

template <int C, int SUB>
__global__ __launch_bounds__(512) void convnext_mlp_fwd_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ W1,
                                                               const float* __restrict__ b1, const bf16_t* __restrict__ W2,
                                                               const float* __restrict__ b2, const float* __restrict__ gamma,
                                                               const float* __restrict__ rowscale, int64_t rows_per_group,
                                                               const bf16_t* __restrict__ R, bf16_t* __restrict__ O, int64_t M) {
    using G = MlpGeom<C, SUB>;
    constexpr int HID = G::HID, KK = G::KK, CB = G::CB, SLAB = G::SLAB, STAGE = G::STAGE, PPW = G::PPW, NST = G::NST, NS = MLP_NS;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* const b1s = reinterpret_cast<float*>(smem + G::RING);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5, q = (lane >> 2) & 3;
    const int64_t m0 = (int64_t)blockIdx.x * 256 + wid * 32;

    // ---- B operand of the first product: this wavefront's 32 rows of y2, all C channels, straight from global memory ----
    bf16x8 yf[KK];
    {
        int64_t row = m0 + r;
        row = row < M ? row : M - 1;
        const bf16_t* yp = Y + row * C + 8 * h;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) yf[kk] = *reinterpret_cast<const bf16x8*>(yp + 16 * kk);
    }
    for (int i = tid; i < HID; i += 512) b1s[i] = b1[i];

    // ---- DMA sources: piece pi = wid + 8 i of a stage; 1 KiB = 16 image rows of 64 B, lane l fills (row l >> 2, 16-B chunk l & 3) ----
    const bf16_t* src[PPW];
    int step[PPW], dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pi = wid + 8 * i;
        const int sub = pi / (C / 8), qq = pi % (C / 8);
        dst[i] = sub * SLAB + qq * 1024;
        if (qq < C / 16) {      // W1 image rows c = 16 qq .. + 15
            src[i] = W1 + (int64_t)(16 * qq + (lane >> 2)) * HID + 32 * sub + 8 * (lane & 3);
            step[i] = 32 * SUB;
        } else {                // W2 image rows R = (cb, hid): piece covers half a 32x32 block
            const int Rr = 16 * (qq - C / 16) + (lane >> 2);
            src[i] = W2 + (int64_t)(32 * sub + (Rr & 31)) * C + 32 * (Rr >> 5) + 8 * (lane & 3);
            step[i] = 32 * SUB * C;
        }
    }
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void_ptr)src[i], (lds_void_ptr)(smem + stage * STAGE + dst[i]), 16, 0, 0);
            src[i] += step[i];
        }
    };

    f32x16 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[cb][j] = 0.f;

    const int colb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int base1 = (8 * h + q) * 64 + colb;      // first product: k rows 16 kk + 8 h + {q, q + 4}
    const int base2 = (4 * h + q) * 64 + colb;      // second product: k rows 16 s + 4 h + {q, q + 8}  (the accumulator's register order)

    auto compute = [&](int stage, int kt) {
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            const char* img1 = smem + stage * STAGE + sub * SLAB;
            const char* img2 = img1 + C * 64;
            f32x16 hacc;
            {
                const float* bb = b1s + 32 * (kt * SUB + sub) + 4 * h;      // register 4 i + u is hidden row 8 i + 4 h + u
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(bb + 8 * i);
                    hacc[4 * i] = v.x;
                    hacc[4 * i + 1] = v.y;
                    hacc[4 * i + 2] = v.z;
                    hacc[4 * i + 3] = v.w;
                }
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(img1 + kk * 1024 + base1, 256), yf[kk], hacc, 0, 0, 0);
            bf16x8 gf[2];
#pragma unroll
            for (int j = 0; j < 16; ++j) gf[j >> 3][j & 7] = (bf16_t)gelu_sig(hacc[j]);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(img2 + cb * 2048 + s * 1024 + base2, 512), gf[s], acc[cb], 0, 0, 0);
        }
    };

    // ---- ring: NS - 1 stages in flight, one barrier per stage (gemm_dma.h) ----
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < NST) issue(p);
    int stage = 0, fill = NS - 1;
    for (int kt = 0; kt < NST; ++kt) {
        if (NST - 1 - kt >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < NST) issue(fill);
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
            float res[16];
            load8<bf16_t>(R + m * C + c0, res);
            load8<bf16_t>(R + m * C + c0 + 8, res + 8);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                float t = v[u] + b2[c0 + u];
                if (gamma) t *= gamma[c0 + u];
                v[u] = fmaf(t, rs, res[u]);
            }
            store8<bf16_t>(O + m * C + c0, v);
            store8<bf16_t>(O + m * C + c0 + 8, v + 8);
        }
    }
}

template <int C, int SUB>
int launch_mlp_fwd(const void* y2, const void* W1, const float* b1, const void* W2, const float* b2, const float* gamma, const float* rowscale,
                   int64_t rows_per_group, const void* residual, void* out, int64_t M, hipStream_t s) {
    using G = MlpGeom<C, SUB>;
    static const bool raised = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&convnext_mlp_fwd_kernel<C, SUB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   G::LDS) == hipSuccess;
    }();
    (void)raised;
    const int grid = (int)ceil_div64(M, 256);
    hipLaunchKernelGGL((convnext_mlp_fwd_kernel<C, SUB>), dim3(grid), dim3(512), G::LDS, s, (const bf16_t*)y2, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2,
                       gamma, rowscale, rows_per_group, (const bf16_t*)residual, (bf16_t*)out, M);
    return iseg_check_launch("iseg_convnext_mlp_fwd");
}

}  // namespace

extern "C" int iseg_convnext_mlp_supported(int C, int dtype) { return dtype == ISEG_BF16 && (C == 96 || C == 192) ? 1 : 0; }

extern "C" int iseg_convnext_mlp_fwd(const void* y2, const void* W1, const float* b1, const void* W2, const float* b2, const float* gamma,
                                     const float* rowscale, int64_t rows_per_group, const void* residual, void* out, int64_t M, int C, int dtype,
                                     hipStream_t stream) {
    ISEG_REQUIRE(iseg_convnext_mlp_supported(C, dtype), "iseg_convnext_mlp_fwd: bf16 storage with C = 96 or 192 only (C = %d, dtype = %d)", C, dtype);
    ISEG_REQUIRE(y2 && W1 && b1 && W2 && b2 && residual && out && M > 0, "iseg_convnext_mlp_fwd: null operand or empty problem");
    ISEG_REQUIRE(!rowscale || rows_per_group > 0, "iseg_convnext_mlp_fwd: rowscale needs rows_per_group > 0");
    ISEG_REQUIRE(((uintptr_t)y2 % 16 == 0) && ((uintptr_t)W1 % 16 == 0) && ((uintptr_t)W2 % 16 == 0) && ((uintptr_t)residual % 16 == 0) &&
                     ((uintptr_t)out % 16 == 0) && ((uintptr_t)b1 % 16 == 0),
                 "iseg_convnext_mlp_fwd: operands must be 16-byte aligned");
    if (C == 96) return launch_mlp_fwd<96, 2>(y2, W1, b1, W2, b2, gamma, rowscale, rows_per_group, residual, out, M, stream);
    return launch_mlp_fwd<192, 1>(y2, W1, b1, W2, b2, gamma, rowscale, rows_per_group, residual, out, M, stream);
}
