// bf16 MFMA GEMM instantiations, orientation "nt" (see gemm_impl.h); split from gemm.hip for parallel compilation.
#include "gemm_impl.h"

namespace iseg_mm {
void gemm_bf16_nt(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    if (g->out_dtype == ISEG_BF16) dispatch_bk<true, true, bf16_t>(g, epi, nsplit, kps, slabs, s);
    else dispatch_bk<true, true, float>(g, epi, nsplit, kps, slabs, s);
}
}  // namespace iseg_mm
