// bf16 MFMA GEMM instantiations, orientation "tn" (see gemm_impl.h); split from gemm.hip for parallel compilation.
#include "gemm_impl.h"

namespace iseg_mm {
void gemm_bf16_tn(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    if (g->out_dtype == ISEG_BF16) dispatch_bk<false, false, bf16_t>(g, epi, nsplit, kps, slabs, s);
    else dispatch_bk<false, false, float>(g, epi, nsplit, kps, slabs, s);
}
}  // namespace iseg_mm
