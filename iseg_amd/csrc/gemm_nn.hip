// bf16 MFMA GEMM instantiations, orientation "nn" (see gemm_impl.h); split from gemm.hip for parallel compilation.
#include "gemm_impl.h"

namespace iseg_mm {
void gemm_bf16_nn(const iseg_gemm_args* g, const Epi& epi, int nsplit, int64_t kps, float* slabs, hipStream_t s) {
    if (g->out_dtype == ISEG_BF16) dispatch_bk<true, false, bf16_t>(g, epi, nsplit, kps, slabs, s);
    else dispatch_bk<true, false, float>(g, epi, nsplit, kps, slabs, s);
}
}  // namespace iseg_mm
