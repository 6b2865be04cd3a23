// bf16 MFMA GEMM instantiations, orientation "tn" (see gemm_impl.h); split from gemm.hip for parallel compilation.
// Split weight-gradient problems that qualify (gemm_dma_tn.h: aligned, M and N >= 128, K >= 2048) take the LDS-DMA pipeline.
#include "gemm_dma_tn.h"

namespace iseg_mm {
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
