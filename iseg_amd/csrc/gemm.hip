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

#include "gemm_dma_tn.h"

using namespace iseg_mm;

#include <stdlib.h>
#include <map>
#include <mutex>
#include <string>
namespace {
// ISEG_GEMM_LOG=1: census of the problems iseg_gemm was asked for (orientation, shape, split, LDS-DMA form, epilogue), printed to stderr at
// exit with the call count of each -- how a configuration's contractions map onto the kernels (tools/prof_config.py shows the time per kernel)
struct GemmCensus {
    std::mutex mu;
    std::map<std::string, long> seen;
    ~GemmCensus() {
        for (const auto& kv : seen) fprintf(stderr, "[iseg_gemm] %6ld x %s\n", kv.second, kv.first.c_str());
    }
};
bool gemm_log_on() {
    static const bool v = [] { const char* e = getenv("ISEG_GEMM_LOG"); return e && atoi(e) != 0; }();
    return v;
}
void gemm_log(const iseg_gemm_args* g, int nsplit) {
    static GemmCensus census;
    char buf[256];
    snprintf(buf, sizeof buf, "%s %s->%s M=%lld N=%lld K=%lld batch=%d split=%d form=%d act=%d a_act=%d%s%s%s%s%s%s%s%s%s%s", g->a_kcontig ? (g->b_kcontig ? "NT" : "NN") : (g->b_kcontig ? "TT" : "TN"),
             g->in_dtype == ISEG_BF16 ? "bf16" : "f32", g->out_dtype == ISEG_BF16 ? "bf16" : "f32", (long long)g->M, (long long)g->N, (long long)g->K,
             g->batch > 1 ? g->batch : 1, nsplit, iseg_gemm_variant(g), g->act, g->a_act, g->bias ? " bias" : "", g->residual ? " residual" : "",
             g->aux ? " aux" : "", g->pre_out ? " pre_out" : "", g->colsum_out ? " colsum" : "", g->colscale ? " colscale" : "", g->rowscale ? " rowscale" : "",
             g->pre_deriv ? " pre_deriv" : "", g->accumulate ? " accumulate" : "", g->alpha != 1.f ? " alpha" : "");
    std::lock_guard<std::mutex> lock(census.mu);
    ++census.seen[buf];
}
}  // namespace
namespace iseg_mm {
int long_k_tile() {
    static const int v = [] {
        const char* e = getenv("ISEG_GEMM_BK");
        return (e && atoi(e) == 64) ? 64 : 128;
    }();
    return v;
}
int dma_mode() {
    static const int v = [] {
        const char* e = getenv("ISEG_GEMM_DMA");
        return e ? atoi(e) : 1;
    }();
    return v;
}
int dma_bk32() {
    static const int v = [] {
        const char* e = getenv("ISEG_GEMM_DMA_BK32");
        return e ? atoi(e) : 1;
    }();
    return v;
}
int dma_min_k() {
    static const int v = [] {
        const char* e = getenv("ISEG_GEMM_DMA_MIN_K");
        const int k = e ? atoi(e) : 32;
        return k < 16 ? 16 : k;
    }();
    return v;
}
int dma_tn_mode() {
    static const int v = [] {
        const char* e = getenv("ISEG_GEMM_DMA_TN");
        return e ? atoi(e) : 1;
    }();
    return v;
}
int tile_waves() {
    static const int v = [] {
        const char* e = getenv("ISEG_GEMM_WAVES");
        const int w = e ? atoi(e) : 8;
        return (w == 4 || w == 16) ? w : 8;
    }();
    return v;
}
}  // namespace iseg_mm

namespace {

// ------------------------------------------------------------------------------------------------
// fp32 kernel (parity path): 64x64 tile, 16-deep k step, 4x4 outputs per thread, k-ordered fmaf
// ------------------------------------------------------------------------------------------------
template <class TO>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                       const float* __restrict__ B, int64_t sbk, int64_t sbn,
                                                       TO* __restrict__ D, int64_t ldd, int64_t M, int64_t N, int64_t K,
                                                       int tiles_n, int64_t k_per_split, float* __restrict__ slabs, Epi epi,
                                                       int a_act) {
    constexpr int TB = 64, TK = 16;
    __shared__ float sA[TK][TB + 1];
    __shared__ float sB[TK][TB + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    if (gridDim.z > 1) {   // strided batch
        A += epi.off_a(blockIdx.z);
        B += epi.off_b(blockIdx.z);
        D += epi.off_d(blockIdx.z);
    }
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
            float av = (m < M && k < kend) ? A[m * sam + k * sak] : 0.f;
            if (a_act == ISEG_ACT_GELU) av = gelu_erf(av);
            sA[ka][ma] = av;
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
                                     int64_t N, Epi epi, int64_t slab_rows) {
    const int64_t total = slab_rows * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += slabs[(int64_t)z * total + i];
        const int64_t m = i / N, n = i % N;
        if (m < M) D[m * ldd + n] = from_f32<TO>(epi_apply<TO>(epi, s, m, n, D, ldd));
        else epi.colsum_out[n] = s + (epi.colsum_accumulate ? epi.colsum_out[n] : 0.f);   // the virtual ones-row
    }
}

// choose the split so that a skinny-output contraction (wgrad: M,N small, K = pixels) still fills 256 CUs
int choose_split(const iseg_gemm_args* g, int tile) {
    if (g->split_k > 0) return g->split_k;
    // weight-gradient orientation on the LDS-DMA pipeline (gemm_dma_tn.h): 256-row tiles, one workgroup per CU
    if (const int form = iseg_mm::dma_tn_form(g)) {
        const int s = iseg_mm::dma_tn_split(g, form);
        if (s > 1) return s;
    }
    const int64_t tiles = ceil_div64(g->M + (g->colsum_out ? 1 : 0), tile) * ceil_div64(g->N, tile);
    if (tiles >= 256 || g->K < 2048 || g->batch > 1 || g->b_group_rows > 0) return 1;
    // the LDS-DMA pipeline keeps several K-tiles in flight per workgroup: half-filled grids are better left unsplit
    if (g->in_dtype == ISEG_BF16 && tiles >= 96 && iseg_mm::dma_mode() && iseg_mm::dma_eligible(g, 128)) return 1;
    // two 128x128 workgroups fit a CU: the grid must stay within ONE resident round of 512 (measured: 42 splits of a 12-tile
    // weight gradient = 504 workgroups 61.5 us, 43 splits = 516 workgroups 74.7 us)
    int64_t want = 512 / tiles;
    if (want < 1) want = 1;
    const int64_t maxs = g->K / 512 > 0 ? g->K / 512 : 1;
    if (want > maxs) want = maxs;
    if (want > 512) want = 512;
    return (int)(want < 1 ? 1 : want);
}

}  // namespace

int gemm_reduce(const iseg_gemm_args* g, const Epi& epi, float* slabs, int eff_split, int64_t slab_rows, hipStream_t stream);

extern "C" int iseg_gemm_splits(const iseg_gemm_args* g) {
    return choose_split(g, g->in_dtype == ISEG_BF16 ? 128 : 64);
}

// slabs a split problem really writes: the K range is cut into 128-element multiples, so it can be fewer than iseg_gemm_splits
extern "C" int iseg_gemm_slabs(const iseg_gemm_args* g) {
    if (!g) return 0;
    const int nsplit = iseg_gemm_splits(g);
    if (nsplit <= 1) return 1;
    const int64_t kps = ceil_div64(ceil_div64(g->K, nsplit), 128) * 128;
    return (int)ceil_div64(g->K, kps);
}

extern "C" int iseg_gemm_variant(const iseg_gemm_args* g) {
    if (!g || g->in_dtype != ISEG_BF16) return 0;
    const int nsplit = iseg_gemm_splits(g);
    const int64_t kps = nsplit > 1 ? ceil_div64(ceil_div64(g->K, nsplit), 128) * 128 : g->K;
    if (nsplit > 1 && iseg_mm::dma_tn_form(g)) return iseg_mm::dma_tn_form(g);      // 7 / 8: the weight-gradient LDS-DMA kernel
    if (!iseg_mm::dma_mode() || !iseg_mm::dma_eligible(g, kps)) return 0;
    return iseg_mm::dma_form(g, (int)ceil_div64(g->K, kps));
}

extern "C" size_t iseg_gemm_workspace_bytes(const iseg_gemm_args* g) {
    const int s = iseg_gemm_splits(g);
    return s > 1 ? (size_t)s * (size_t)(g->M + (g->colsum_out ? 1 : 0)) * (size_t)g->N * sizeof(float) : 0;
}

extern "C" int iseg_gemm(const iseg_gemm_args* g, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(g && g->A && g->B && g->D, "iseg_gemm: null operand");
    ISEG_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0, "iseg_gemm: empty problem M=%lld N=%lld K=%lld", (long long)g->M,
                 (long long)g->N, (long long)g->K);
    ISEG_REQUIRE(g->in_dtype == ISEG_F32 || g->in_dtype == ISEG_BF16, "iseg_gemm: bad in_dtype %d", g->in_dtype);
    ISEG_REQUIRE(g->out_dtype == ISEG_F32 || g->out_dtype == ISEG_BF16, "iseg_gemm: bad out_dtype %d", g->out_dtype);
    ISEG_REQUIRE(!(g->in_dtype == ISEG_F32 && g->out_dtype == ISEG_BF16), "iseg_gemm: f32 inputs need f32 output");
    ISEG_REQUIRE(!g->rowscale || g->rows_per_group > 0, "iseg_gemm: rowscale needs rows_per_group");
    ISEG_REQUIRE((g->act != ISEG_ACT_GELU_GRAD && g->act != ISEG_ACT_RELU_GRAD && g->act != ISEG_ACT_MUL_AUX) || g->aux,
                 "iseg_gemm: act needs aux");
    ISEG_REQUIRE(g->act >= ISEG_ACT_NONE && g->act <= ISEG_ACT_MUL_AUX, "iseg_gemm: bad act %d", g->act);
    ISEG_REQUIRE(!g->pre_deriv || (g->act == ISEG_ACT_GELU && g->pre_out), "iseg_gemm: pre_deriv needs act = GELU and pre_out");
    ISEG_REQUIRE(!g->bias_rowscaled || (g->bias && g->rowscale), "iseg_gemm: bias_rowscaled needs bias and rowscale");
    ISEG_REQUIRE(g->a_act == ISEG_ACT_NONE || g->a_act == ISEG_ACT_GELU, "iseg_gemm: a_act must be NONE or GELU");
    Epi epi{g->bias, g->colscale, g->rowscale, g->rows_per_group, g->residual, g->ldr, g->aux, g->ldaux, g->pre_out, g->ldp,
            g->act, g->alpha, g->accumulate, g->colsum_out, g->colsum_accumulate,
            g->batch_inner > 0 ? g->batch_inner : 1, g->sa_outer, g->sa_inner, g->sb_outer, g->sb_inner, g->sd_outer, g->sd_inner,
            g->pre_deriv, g->b_group_rows, g->b_group_stride, g->bias_rowscaled};
    const int batch = g->batch > 1 ? g->batch : 1;
    if (batch > 1) {
        ISEG_REQUIRE(batch <= 65535, "iseg_gemm: batch %d exceeds the grid z limit (65535)", batch);
        ISEG_REQUIRE(!g->bias && !g->colscale && !g->rowscale && !g->residual && !g->aux && !g->pre_out && !g->colsum_out &&
                         g->act == ISEG_ACT_NONE && g->a_act == ISEG_ACT_NONE,
                     "iseg_gemm: a batched problem takes only the alpha / accumulate epilogue");
    }
    if (g->b_group_rows > 0) {
        if (!(g->in_dtype == ISEG_BF16 && g->a_kcontig && g->b_kcontig && batch == 1 && g->split_k <= 1 && g->b_group_rows % 256 == 0 &&
              g->b_group_stride % 8 == 0 && iseg_mm::dma_mode() && iseg_mm::dma_eligible(g, g->K))) {
            iseg_set_error("iseg_gemm: B per row group needs the LDS-DMA path (bf16, K-contiguous operands, K %% 64 == 0, aligned) with "
                           "b_group_rows %% 256 == 0, no split and no batch");
            return ISEG_ERR_UNSUPPORTED;
        }
    }
    if (g->colsum_out) {
        ISEG_REQUIRE(g->in_dtype == ISEG_BF16 && !g->a_kcontig && !g->b_kcontig, "iseg_gemm: colsum_out needs the bf16 wgrad orientation");
        ISEG_REQUIRE(g->M % 8 == 0, "iseg_gemm: colsum_out needs M %% 8 == 0");      // (M %% 128 == 0: the ones-row opens one more tile row)
    }
    const int64_t slab_rows = g->M + (g->colsum_out ? 1 : 0);
    const int nsplit = iseg_gemm_splits(g);
    if (gemm_log_on()) gemm_log(g, nsplit);
    float* slabs = nullptr;
    int64_t kps = g->K;
    if (nsplit > 1) {
        const size_t need = (size_t)nsplit * slab_rows * g->N * sizeof(float);
        if (!ws || ws_bytes < need) {
            iseg_set_error("iseg_gemm: split-K needs %zu workspace bytes, got %zu", need, ws_bytes);
            return ISEG_ERR_WORKSPACE;
        }
        slabs = (float*)ws;
        kps = ceil_div64(ceil_div64(g->K, nsplit), 128) * 128;
    }
    const int eff_split = (int)ceil_div64(g->K, kps);
    if (g->in_dtype == ISEG_BF16) {
        if (g->a_kcontig && !g->b_kcontig) gemm_bf16_nn(g, epi, eff_split, kps, slabs, stream);
        else if (g->a_kcontig && g->b_kcontig) gemm_bf16_nt(g, epi, eff_split, kps, slabs, stream);
        else if (!g->a_kcontig && !g->b_kcontig) gemm_bf16_tn(g, epi, eff_split, kps, slabs, stream);
        else {
            iseg_set_error("iseg_gemm: bf16 path does not provide A MN-contiguous x B K-contiguous");
            return ISEG_ERR_UNSUPPORTED;
        }
    } else {
        const int tiles_m = (int)ceil_div64(g->M, 64), tiles_n = (int)ceil_div64(g->N, 64);
        const int64_t sam = g->a_kcontig ? g->lda : 1, sak = g->a_kcontig ? 1 : g->lda;
        const int64_t sbk = g->b_kcontig ? 1 : g->ldb, sbn = g->b_kcontig ? g->ldb : 1;
        hipLaunchKernelGGL((gemm_f32_kernel<float>), dim3(tiles_m * tiles_n, eff_split, batch), dim3(256), 0, stream,
                           (const float*)g->A, sam, sak, (const float*)g->B, sbk, sbn, (float*)g->D, g->ldd, g->M, g->N, g->K,
                           tiles_n, kps, slabs, epi, g->a_act);
    }
    if (g->defer_reduce) return iseg_check_launch("iseg_gemm");   // caller runs iseg_gemm_reduce itself
    return gemm_reduce(g, epi, slabs, eff_split, slab_rows, stream);
}

int gemm_reduce(const iseg_gemm_args* g, const Epi& epi, float* slabs, int eff_split, int64_t slab_rows, hipStream_t stream) {
    const bool plain_epilogue = !g->bias && !g->colscale && !g->rowscale && !g->residual && !g->pre_out && g->act == ISEG_ACT_NONE &&
                                g->out_dtype == ISEG_F32 && g->ldd == g->N &&
                                (!g->colsum_out || (g->colsum_accumulate != 0) == (g->accumulate != 0));
    if (slabs && plain_epilogue) {
        // weight gradients: D (+)= alpha * sum_z slab[z] (and the ones-row -> colsum_out) with the 2-D parallel reducer
        const int64_t mn = g->M * g->N;
        launch_reduce_rows(slabs, eff_split, slab_rows * g->N, 0, 1, slab_rows * g->N, (float*)g->D, g->colsum_out, mn, 0, g->alpha,
                           g->accumulate, stream);
        (void)mn;
    } else if (slabs) {
        const int64_t total = slab_rows * g->N;
        const int blocks = (int)(ceil_div64(total, 256) < 2048 ? ceil_div64(total, 256) : 2048);
        if (g->out_dtype == ISEG_BF16)
            hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, slabs, eff_split,
                               (bf16_t*)g->D, g->ldd, g->M, g->N, epi, slab_rows);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(blocks), dim3(256), 0, stream, slabs, eff_split,
                               (float*)g->D, g->ldd, g->M, g->N, epi, slab_rows);
    }
    return iseg_check_launch("iseg_gemm");
}

// Two weight-gradient problems over the same reduction rows as ONE launch (gemm_dma_tn.h, gemm_bf16_dma_tn_pair_kernel).
//   splits = iseg_gemm_tn_pair_splits(g0, g1)    0: do not pair (run them one by one)
//   the caller sets split_k = splits in both argument blocks, sizes their workspaces with iseg_gemm_workspace_bytes, calls iseg_gemm_tn_pair (which
//   only writes the slabs, like defer_reduce = 1) and finishes each problem with iseg_gemm_reduce or a consumer of the slabs.
extern "C" int iseg_gemm_tn_pair_splits(const iseg_gemm_args* g0, const iseg_gemm_args* g1) {
    if (!g0 || !g1) return 0;
    iseg_gemm_args a = *g0, b = *g1;
    a.split_k = b.split_k = 0;
    static const int off = [] { const char* e = getenv("ISEG_GEMM_TN_PAIR"); return e && atoi(e) == 0; }();
    return off ? 0 : iseg_mm::gemm_bf16_tn_pair_split(&a, &b);
}

extern "C" int iseg_gemm_tn_pair(const iseg_gemm_args* g0, void* ws0, size_t ws0_bytes, const iseg_gemm_args* g1, void* ws1, size_t ws1_bytes,
                                 hipStream_t stream) {
    ISEG_REQUIRE(g0 && g1 && g0->A && g0->B && g1->A && g1->B, "iseg_gemm_tn_pair: null operand");
    const int s = iseg_gemm_tn_pair_splits(g0, g1);
    ISEG_REQUIRE(s > 1 && g0->split_k == s && g1->split_k == s, "iseg_gemm_tn_pair: the problems do not pair (splits %d, split_k %d / %d)", s, g0->split_k,
                 g1->split_k);
    ISEG_REQUIRE(!g0->bias && !g1->bias && g0->act == ISEG_ACT_NONE && g1->act == ISEG_ACT_NONE && g0->out_dtype == ISEG_F32 && g1->out_dtype == ISEG_F32,
                 "iseg_gemm_tn_pair: plain fp32 weight gradients only");
    const size_t need0 = iseg_gemm_workspace_bytes(g0), need1 = iseg_gemm_workspace_bytes(g1);
    if (!ws0 || ws0_bytes < need0 || !ws1 || ws1_bytes < need1) {
        iseg_set_error("iseg_gemm_tn_pair: needs %zu + %zu workspace bytes, got %zu + %zu", need0, need1, ws0_bytes, ws1_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const int64_t kps = ceil_div64(ceil_div64(g0->K, s), 128) * 128;
    const int eff = (int)ceil_div64(g0->K, kps);
    iseg_mm::gemm_bf16_tn_pair(g0, (float*)ws0, g1, (float*)ws1, eff, kps, stream);
    return iseg_check_launch("iseg_gemm_tn_pair");
}

// second half of a deferred split-K GEMM (args.defer_reduce = 1): slab-order sum + epilogue.  No-op when the problem is not split.
extern "C" int iseg_gemm_reduce(const iseg_gemm_args* g, void* ws, size_t ws_bytes, hipStream_t stream) {
    ISEG_REQUIRE(g && g->D, "iseg_gemm_reduce: null argument");
    const int nsplit = iseg_gemm_splits(g);
    if (nsplit <= 1) return ISEG_OK;
    const int64_t slab_rows = g->M + (g->colsum_out ? 1 : 0);
    const size_t need = (size_t)nsplit * slab_rows * g->N * sizeof(float);
    if (!ws || ws_bytes < need) {
        iseg_set_error("iseg_gemm_reduce: needs %zu workspace bytes, got %zu", need, ws_bytes);
        return ISEG_ERR_WORKSPACE;
    }
    const int64_t kps = ceil_div64(ceil_div64(g->K, nsplit), 128) * 128;
    const int eff_split = (int)ceil_div64(g->K, kps);
    Epi epi{g->bias, g->colscale, g->rowscale, g->rows_per_group, g->residual, g->ldr, g->aux, g->ldaux, g->pre_out, g->ldp,
            g->act, g->alpha, g->accumulate, g->colsum_out, g->colsum_accumulate,
            g->batch_inner > 0 ? g->batch_inner : 1, g->sa_outer, g->sa_inner, g->sb_outer, g->sb_inner, g->sd_outer, g->sd_inner,
            g->pre_deriv, g->b_group_rows, g->b_group_stride, g->bias_rowscaled};
    const int batch = g->batch > 1 ? g->batch : 1;
    if (batch > 1) {
        ISEG_REQUIRE(batch <= 65535, "iseg_gemm: batch %d exceeds the grid z limit (65535)", batch);
        ISEG_REQUIRE(!g->bias && !g->colscale && !g->rowscale && !g->residual && !g->aux && !g->pre_out && !g->colsum_out &&
                         g->act == ISEG_ACT_NONE && g->a_act == ISEG_ACT_NONE,
                     "iseg_gemm: a batched problem takes only the alpha / accumulate epilogue");
    }
    return gemm_reduce(g, epi, (float*)ws, eff_split, slab_rows, stream);
}
