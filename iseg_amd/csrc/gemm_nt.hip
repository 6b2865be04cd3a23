// bf16 MFMA GEMM instantiations, orientation "nt" (see gemm_impl.h); split from gemm.hip for parallel compilation.
// Eligible problems (gemm_dma.h: aligned, K % 64 == 0, no operand transform) take the LDS-DMA pipeline.
#include "gemm_dma.h"

namespace iseg_mm {
void gemm_bf16_nt(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    if (dma_mode() && dma_eligible(g, kps)) {
        if (g->out_dtype == ISEG_BF16) dispatch_dma<bf16_t>(g, epi, nsplit, kps, slabs, s);
        else dispatch_dma<float>(g, epi, nsplit, kps, slabs, s);
        return;
    }
    if (g->out_dtype == ISEG_BF16) dispatch_bk<true, true, bf16_t>(g, epi, nsplit, kps, slabs, s);
    else dispatch_bk<true, true, float>(g, epi, nsplit, kps, slabs, s);
}
}  // namespace iseg_mm
