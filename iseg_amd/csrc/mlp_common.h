// Helpers shared by the fused ConvNeXt MLP kernels (mlp_fused.hip, mlp_wgrad.hip).
#pragma once
#include "common.h"

// LayerNormalization(axis = -1) applied to one element with the row's saved statistics -- the expression of layernorm_fwd_kernel
// (norm.hip), so that a kernel which re-forms y2 = LN(y1) while staging its operand sees the values the forward pass used.
__device__ __forceinline__ float iseg_ln_apply(float x, float mean, float rstd, float g, float b) { return (x - mean) * rstd * g + b; }
