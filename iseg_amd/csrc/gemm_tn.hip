// bf16 MFMA GEMM instantiations, orientation "tn" (see gemm_impl.h); split from gemm.hip for parallel compilation.
// Split weight-gradient problems that qualify (gemm_dma_tn.h: aligned, M and N >= 128, K >= 2048) take the LDS-DMA pipeline.
#include "gemm_dma_tn.h"

namespace iseg_mm {
bool dma_tn_lds_ok() {
    static const bool ok = [] {
        constexpr int lds = 3 * 64 * (256 + 128) * 2;      // three 64-row stages of a 256 x 128 / 128 x 256 tile pair
        const bool a = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_tn_kernel<4, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        const bool b = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_tn_kernel<2, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        const bool c = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_tn_pair_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        if (!(a && b && c)) (void)hipGetLastError();      // not an error of the call that asked: the register-staged kernel takes these problems
        return a && b && c;
    }();
    return ok;
}

int gemm_bf16_tn_pair_split(const iseg_gemm_args* g0, const iseg_gemm_args* g1) { return dma_tn_pair_split(g0, g1); }
void gemm_bf16_tn_pair(const iseg_gemm_args* g0, float* slabs0, const iseg_gemm_args* g1, float* slabs1, int nsplit, int64_t kps, hipStream_t s) {
    launch_dma_tn_pair(g0, slabs0, g1, slabs1, nsplit, kps, s);
}

void gemm_bf16_tn(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    if (slabs && nsplit > 1 && kps % 64 == 0) {
        const int form = dma_tn_form(g);
        if (form == 7) return launch_dma_tn<4, 2>(g, nsplit, kps, slabs, s);
        if (form == 8) return launch_dma_tn<2, 4>(g, nsplit, kps, slabs, s);
    }
    if (g->out_dtype == ISEG_BF16) dispatch_bk<false, false, bf16_t>(g, epi, nsplit, kps, slabs, s);
    else dispatch_bk<false, false, float>(g, epi, nsplit, kps, slabs, s);
}
}  // namespace iseg_mm
