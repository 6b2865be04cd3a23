/* iseg_hip.h -- C ABI of libiseg_hip.so, the MI355X (gfx950) compute library behind iseg_amd.
 *
 * The reference (edwardyehuang/iSeg) has no native boundary: its "kernels" are TensorFlow/Keras ops reached
 * through Python call signatures.  Each entry point below therefore cites the reference Python call site(s)
 * whose per-step arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - C linkage, plain pointers and sizes; no torch / C++ types.
 *   - Pointers are DEVICE pointers unless the name ends in _h. Tensors are dense NHWC / row-major.
 *   - dtype: ISEG_F32 (0) or ISEG_BF16 (1) storage; arithmetic is fp32 (MFMA accumulates in fp32).
 *   - Returns ISEG_OK (0) or a negative iseg status; iseg_last_error() holds a per-thread message.
 *   - Nothing allocates or frees device memory: the caller owns inputs, outputs and workspace
 *     (query iseg_*_workspace_bytes first).  All work is enqueued on `stream`; entry points never
 *     synchronise, so they can be captured into a hipGraph.
 *   - Keras weight layouts are consumed as they are: Dense [in,out], Conv2D [kh,kw,Cin,Cout],
 *     DepthwiseConv2D [kh,kw,C,1].
 */
#ifndef ISEG_HIP_H
#define ISEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ISEG_HIP_STREAM_T
#define ISEG_HIP_STREAM_T
typedef struct ihipStream_t* iseg_stream_t; /* == hipStream_t */
#endif

#define ISEG_STATUS_OK 0
#define ISEG_STATUS_ARG (-1)
#define ISEG_STATUS_HIP (-2)
#define ISEG_STATUS_UNSUPPORTED (-3)
#define ISEG_STATUS_WORKSPACE (-4)

#define ISEG_DTYPE_F32 0
#define ISEG_DTYPE_BF16 1

/* epilogue activations of iseg_gemm */
#define ISEG_ACT_NONE 0
#define ISEG_ACT_RELU 1      /* keras.activations.relu   (layers/model_builder.py:66,92-93) */
#define ISEG_ACT_GELU 2      /* keras.activations.gelu, exact erf (backbones/convnext.py:53) */
#define ISEG_ACT_GELU_GRAD 3 /* v *= gelu'(aux)  : backward of ISEG_ACT_GELU, aux = saved pre-activation */
#define ISEG_ACT_RELU_GRAD 4 /* v  = aux > 0 ? v : 0 */
#define ISEG_ACT_MUL_AUX 5   /* v *= aux : backward of ISEG_ACT_GELU when the forward saved gelu'(pre) (pre_deriv = 1) */
#define ISEG_ACT_SIGMOID 6   /* tf.nn.sigmoid (layers/nasfpn.py:304-311 global-attention gate); iseg_act_fwd / iseg_act_bwd only */
#define ISEG_ACT_SWISH 7     /* tf.nn.silu / keras "swish": x sigmoid(x) (backbones/eva/swiglu.py:13, layers/nasfpn.py:289); iseg_act_fwd / _bwd only */

int iseg_version(void);
/* copies the calling thread's last error message (NUL-terminated) into buf_h; returns its length */
size_t iseg_last_error(char* buf_h, size_t n);

/* ---------------------------------------------------------------------------------------------------------
 * Dense contraction  D[M,N] = epilogue( alpha * sum_k A(m,k) B(k,n) )
 * Replaces: keras.layers.Dense (backbones/convnext.py:29-30,51-54; backbones/swin.py Mlp/qkv/proj;
 * backbones/vit.py MLPBlock), 1x1 keras.layers.Conv2D (layers/model_builder.py:54-64; layers/core_model_ext.py:129)
 * and, after iseg_im2col, every spatial Conv2D (layers/aspp.py:41-52; backbones/convnext.py:72-75), plus their
 * autodiff transposes (dgrad: a_kcontig=1,b_kcontig=1 on the Keras kernel as stored; wgrad: a_kcontig=0,b_kcontig=0).
 *   a_kcontig=1: A(m,k) = A[m*lda + k]      a_kcontig=0: A(m,k) = A[k*lda + m]
 *   b_kcontig=1: B(k,n) = B[n*ldb + k]      b_kcontig=0: B(k,n) = B[k*ldb + n]   (Keras [in,out])
 * Epilogue, in this order:  v = alpha*acc ; v += bias[n] ; pre_out[m,n] = v (or gelu'(v), pre_deriv) ; act ; v *= colscale[n] ;
 *   v *= rowscale[m / rows_per_group] ; v += residual[m,n] ; v += D[m,n] if accumulate ; D[m,n] = v.
 * residual / aux / pre_out have the OUTPUT dtype.  in_dtype bf16 -> MFMA path; f32 -> fp32 FMA (parity) path.
 * split_k: 0 = automatic (skinny outputs with a long reduction are split and reduced in slab order:
 * deterministic), n>0 = forced.  Workspace: iseg_gemm_workspace_bytes().
 * --------------------------------------------------------------------------------------------------------- */
typedef struct iseg_gemm_args {
    const void* A;
    int64_t lda;
    int a_kcontig;
    const void* B;
    int64_t ldb;
    int b_kcontig;
    void* D;
    int64_t ldd;
    int64_t M, N, K;
    int in_dtype, out_dtype;
    const float* bias;     /* [N] or NULL */
    const float* colscale; /* [N] or NULL  (ConvNeXt layer scale gamma, backbones/convnext.py:56-57) */
    const float* rowscale; /* [ceil(M/rows_per_group)] or NULL (drop_path per-sample factor, utils/drops.py:8-22) */
    int64_t rows_per_group;
    const void* residual; /* [M, ldr] or NULL */
    int64_t ldr;
    const void* aux; /* [M, ldaux], needed by ISEG_ACT_*_GRAD and ISEG_ACT_MUL_AUX */
    int64_t ldaux;
    void* pre_out; /* [M, ldp] or NULL: pre-activation saved for backward */
    int64_t ldp;
    int act;
    float alpha;
    int accumulate;
    int split_k;
    int a_act; /* ISEG_ACT_NONE or ISEG_ACT_GELU: A := gelu(A) applied while the operand is staged (the GELU output of
                  backbones/convnext.py:53 is never materialised; pwconv2 and its weight gradient re-derive it) */
    float* colsum_out; /* [N] or NULL.  wgrad orientation only (a_kcontig=0,b_kcontig=0, bf16, M % 8 == 0): also returns
                          sum_k B(k,:) -- the Dense bias gradient -- from a virtual ones-row of A, i.e. without another pass
                          over the [pixels, N] gradient tensor */
    int colsum_accumulate;
    int defer_reduce; /* 1: a split-K problem only writes its slabs; the caller finishes with iseg_gemm_reduce (lets the two
                         kernels be timed / scheduled separately) */
    /* strided batch (attention: one problem per (sample, head)); batch <= 1 = a single problem.  Problem z reads/writes at
       X + (z / batch_inner) * sX_outer + (z % batch_inner) * sX_inner (elements).  Batched problems take only the
       alpha / accumulate epilogue and are never split along K. */
    int batch, batch_inner;
    int64_t sa_outer, sa_inner, sb_outer, sb_inner, sd_outer, sd_inner;
    int pre_deriv; /* 1 (needs act = ISEG_ACT_GELU and pre_out): pre_out receives gelu'(pre-activation) instead of the pre-activation --
                      the forward epilogue has Phi(v) and exp(-v^2/2) in hand anyway, and the backward GEMM's epilogue becomes one
                      multiply (ISEG_ACT_MUL_AUX) instead of re-evaluating erf per element (50 us of VALU per 100 M elements) */
    /* B per row group (b_group_rows > 0): rows [i*b_group_rows, (i+1)*b_group_rows) of A / D are multiplied by B + i*b_group_stride
       (elements) -- one kernel per sample, e.g. the Dense after ConvNeXt V2's response normalisation with the per-sample channel factors
       folded into its kernel (backbones/convnext_v2.py:92-93).  Every epilogue option stays available (rows keep their global index).
       bf16, both operands K-contiguous, b_group_rows % 256 == 0, no split-K, no batch; anything else returns ISEG_ERR_UNSUPPORTED. */
    int64_t b_group_rows, b_group_stride;
    int bias_rowscaled; /* 1 (needs bias and rowscale): the row factor multiplies the BIAS instead of the result, D = (acc + bias * rowscale[m / rows_per_group])
                           * colscale + residual -- the pwconv2 forward of a ConvNeXt block whose A operand already carries the drop-path factor
                           (the pwconv1 epilogue wrote rowscale * gelu(h)): x + s gamma (g W2 + b2) = x + gamma ((s g) W2 + s b2),
                           backbones/convnext.py:56-63.  The scaled activation then serves the weight gradient Z = (s g)^T dout directly, and the
                           backward pass needs no scaled copy of dout. */
} iseg_gemm_args;

int iseg_gemm_splits(const iseg_gemm_args* args_h);
/* slabs a deferred split problem really writes ([slab][M (+1 with colsum_out)][N] fp32 in the workspace): <= iseg_gemm_splits, the K range is cut
 * into multiples of 128 -- for a consumer that sums the slabs itself (iseg_layerscale_grads_slabs) */
int iseg_gemm_slabs(const iseg_gemm_args* args_h);
/* which main loop iseg_gemm runs for this problem (profiling labels): 0 = register-staged gemm_bf16_kernel / fp32 kernel,
   1..4 = LDS-DMA pipeline gemm_bf16_dma_kernel with tile 128x64 / 256x128 / 128x128 (2 stages) / 128x128 (3 stages),
   5 = 256x128 persistent (one workgroup per CU walks several tiles), 6 = 256x192 (2 stages),
   7 / 8 = the weight-gradient orientation (a_kcontig = b_kcontig = 0, split over K: Dense / 1x1-conv kernel gradients,
   layers' `kernel` of backbones/convnext.py:51-55, backbones/swin.py:17-43) on the LDS-DMA pipeline gemm_bf16_dma_tn_kernel with
   256x128 / 128x256 tiles (M, N multiples of 8 and >= 128 -- or one of them 64..127 when the other is >= 320 --, any K >= 2048, aligned operands; ISEG_GEMM_DMA_TN=0 pins the
   register-staged kernel) */
int iseg_gemm_variant(const iseg_gemm_args* args_h);
size_t iseg_gemm_workspace_bytes(const iseg_gemm_args* args_h);
int iseg_gemm(const iseg_gemm_args* args_h, void* ws, size_t ws_bytes, iseg_stream_t stream);
int iseg_gemm_reduce(const iseg_gemm_args* args_h, void* ws, size_t ws_bytes, iseg_stream_t stream);
/* Two weight-gradient problems over the SAME reduction rows in one launch (the pair of an un-fused ConvNeXt block's backward pass,
 * backbones/convnext.py:51-54: Z = gelu(h)^T dout and dW1 = y2^T dH).  iseg_gemm_tn_pair_splits: the common split count, 0 = run them one by one.
 * The caller sets split_k to it in both blocks; iseg_gemm_tn_pair writes both problems' slabs (as defer_reduce = 1 does) and each problem is
 * finished by iseg_gemm_reduce or by a consumer of its slabs (iseg_layerscale_grads_slabs). */
int iseg_gemm_tn_pair_splits(const iseg_gemm_args* g0, const iseg_gemm_args* g1);
int iseg_gemm_tn_pair(const iseg_gemm_args* g0, void* ws0, size_t ws0_bytes, const iseg_gemm_args* g1, void* ws1, size_t ws1_bytes,
                      iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * LayerNorm over the last axis: keras.layers.LayerNormalization(axis=-1, epsilon)
 * (backbones/convnext.py:27,71; backbones/swin.py norm1/norm2; backbones/vit.py). x,y: [rows, C].  C % 8 == 0 takes the 16-byte-vector kernels;
 * any other C (<= 4096 for the backward; EVA02-large's 2730 hidden units, backbones/eva/swiglu.py:74-80) a one-wavefront-per-row form -- plain
 * iseg_layernorm_fwd / _bwd only, not the gather / post-norm entry points.
 * mean/rstd [rows] are saved for the backward.  bwd: dx (= dx_add + LN^T dy), dgamma, dbeta.
 * --------------------------------------------------------------------------------------------------------- */
int iseg_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                       int C, float eps, int dtype, iseg_stream_t stream);
size_t iseg_layernorm_bwd_workspace_bytes(int64_t rows, int C);
int iseg_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                       const void* dx_add, float* dgamma, float* dbeta, int accumulate_param_grads, int64_t rows, int C,
                       int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);
/* Tail of a post-norm residual branch in one pass each way (backbones/intern_image/intern_image.py:226-236: x = residual + drop_path(gamma *
 * norm(f(x))), utils/drops.py:8-22): y = residual + rowscale[row / rows_per_group] * colscale[c] * LN(x); colscale [C], rowscale, residual
 * optional.  Backward: dx = LN^T(rowscale colscale dy); dgamma, dbeta and dcolscale (NULL: not wanted) are ACCUMULATED -- the three come from
 * the same two column sums.  Workspace of the backward: iseg_layernorm_bwd_workspace_bytes(rows, C) + 2 C floats. */
int iseg_layernorm_post_fwd(const void* x, const float* gamma, const float* beta, const float* colscale, const float* rowscale,
                            int64_t rows_per_group, const void* residual, void* y, float* mean, float* rstd, int64_t rows, int C, float eps,
                            int dtype, iseg_stream_t stream);
int iseg_layernorm_post_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* colscale, const float* rowscale,
                            int64_t rows_per_group, const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                            float* dcolscale, int64_t rows, int C, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);
/* LayerNorm followed by a static row permutation with zero padding -- norm1 + tf.pad + tf.roll + window_partition of a Swin block
 * (backbones/swin.py:246-262) in one pass: y[r,:] = src_index[r] >= 0 ? LN(x[src_index[r],:]) : 0 for r < rows_out; mean / rstd [rows_out]
 * are kept per OUTPUT row.  Backward over the `rows` SOURCE rows with the inverse table: the gradient row and the statistics of source row
 * r sit at row dy_index[r] (< 0: no gradient arrives); dx = dx_add + LN^T dy as in iseg_layernorm_bwd (same workspace). */
int iseg_layernorm_gather_fwd(const void* x, const int32_t* src_index, const float* gamma, const float* beta, void* y, float* mean,
                              float* rstd, int64_t rows_out, int C, float eps, int dtype, iseg_stream_t stream);
int iseg_layernorm_gather_bwd(const void* dy, const int32_t* dy_index, const void* x, const float* gamma, const float* mean,
                              const float* rstd, void* dx, const void* dx_add, float* dgamma, float* dbeta, int accumulate_param_grads,
                              int64_t rows, int C, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Depthwise KxK conv, stride 1, dilation `dil`, explicit top/left padding (TF "same": pad = (K-1)*dil/2):
 * keras.layers.DepthwiseConv2D (backbones/convnext.py:25,50; build_dilated_convnext :245-266).
 * w: [K*K, C] fp32 (the Keras [K,K,C,1] kernel), bias [C] or NULL.  y = conv(x) + bias (+ add).
 * flip=1 turns it into the data gradient: call with dy as x, pad' = (K-1)*dil - pad, bias NULL, and
 * `add` = gradient arriving through the residual branch.
 * --------------------------------------------------------------------------------------------------------- */
int iseg_dwconv2d_fwd(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C,
                      int K, int dil, int pad_t, int pad_l, int flip, int dtype, iseg_stream_t stream);
/* The 7 x 7 / bf16 / C % 32 == 0 case of iseg_dwconv2d_fwd on the matrix cores (csrc/dwconv_mfma.hip: banded products with
 * v_mfma_f32_4x4x4_16b_bf16, block = channel) -- the route iseg_dwconv2d_fwd takes by itself for large planes (ISEG_DW_MFMA), exported so that
 * tests and benchmarks can name it.  Same arguments and semantics (backbones/convnext.py:23-27,47-50); weights are rounded to bf16 as the
 * reference's mixed_bfloat16 policy does.  ISEG_ERR_UNSUPPORTED when the shape is not eligible. */
int iseg_dwconv2d7_mfma(const void* x, const float* w, const float* bias, const void* add, void* y, int N, int H, int W, int C, int pad_t,
                        int pad_l, int flip, iseg_stream_t stream);
size_t iseg_dwconv2d_bwd_weight_workspace_bytes(int N, int H, int W, int C, int K);
int iseg_dwconv2d_bwd_weight(const void* x, const void* dy, float* dw, float* db, int accumulate, int N, int H, int W, int C,
                             int K, int dil, int pad_t, int pad_l, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);

/* The same layer with strides = s (backbones/mobilenetv2.py:60-78 the 3x3 / s2 of an inverted residual block; the strided SepConvBnReLU of
 * layers/model_builder.py), computed at the strided output positions only.  pad_t / pad_l: TF "same" front padding for (H, K, s, dil);
 * Ho = ceil(H / s), Wo = ceil(W / s).  The data gradient is the gather over dy (a tap contributes where (h + pad_t - i*dil) is a multiple of
 * s); the weight gradient sums fixed pixel partitions in a fixed order (deterministic) into dw [K*K, C] / db [C] (+= when accumulate). */
int iseg_dwconv2d_strided_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C, int K, int stride, int dil,
                              int pad_t, int pad_l, int Ho, int Wo, int dtype, iseg_stream_t stream);
int iseg_dwconv2d_strided_bwd_data(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int K, int stride, int dil, int pad_t,
                                   int pad_l, int Ho, int Wo, int dtype, iseg_stream_t stream);
size_t iseg_dwconv2d_strided_bwd_weight_workspace_bytes(int N, int Ho, int Wo, int C, int K);
int iseg_dwconv2d_strided_bwd_weight(const void* x, const void* dy, float* dw, float* db, int accumulate, int N, int H, int W, int C, int K,
                                     int stride, int dil, int pad_t, int pad_l, int Ho, int Wo, int dtype, void* ws, size_t ws_bytes,
                                     iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * (Sync)BatchNorm, training mode: keras BatchNormalization(synchronized=True) as wired by
 * layers/normalizations.py:14-23,39-132; moments as layers/keras3/bn.py:10-73 / layers/syncbn.py:70-119.
 *   1. iseg_bn_stats       packed[0:C]=sum x, [C:2C]=sum x^2, [2C]=local count   (one message for RCCL all-reduce)
 *   2. (caller) all-reduce(sum) of packed over ranks
 *   3. iseg_bn_finalize    mean, rstd=rsqrt(var_biased+eps); moving <- moving*momentum + batch*(1-momentum)
 *   4. iseg_bn_apply_fwd   y = act((x-mean)*rstd*gamma+beta); y may be a channel slice (ldy) of a concat buffer
 *  bwd: iseg_bn_bwd_reduce sums[0:C]=sum dz, [C:2C]=sum dz*xhat (dz = dy*relu'(y)); (caller) all-reduce;
 *       iseg_bn_bwd_apply  dx = gamma*rstd*(dz - sums0/n - xhat*sums1/n);  dgamma = sums[C:2C], dbeta = sums[0:C].
 * x: [rows, C] with row stride ldx (elements); C, ld* multiples of 8.
 * --------------------------------------------------------------------------------------------------------- */
size_t iseg_bn_workspace_bytes(int64_t rows, int C);
int iseg_bn_stats(const void* x, int64_t ldx, float* packed, int64_t rows, int C, int dtype, void* ws, size_t ws_bytes,
                  iseg_stream_t stream);
int iseg_bn_finalize(const float* packed, int C, float eps, float momentum, float* mean, float* rstd, float* moving_mean,
                     float* moving_var, iseg_stream_t stream);
int iseg_bn_apply_fwd(const void* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma, const float* beta,
                      void* y, int64_t ldy, int64_t rows, int C, int relu, int dtype, iseg_stream_t stream);
/* iseg_bn_finalize + iseg_bn_apply_fwd in one launch (same arithmetic): mean, rstd [C] are outputs, moving_* updated in place (may be NULL) */
int iseg_bn_apply_fwd_packed(const void* x, int64_t ldx, const float* packed, float eps, float momentum, const float* gamma, const float* beta,
                             float* mean, float* rstd, float* moving_mean, float* moving_var, void* y, int64_t ldy, int64_t rows, int C,
                             int relu, int dtype, iseg_stream_t stream);
int iseg_bn_bwd_reduce(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* y, int64_t ldy, const float* mean,
                       const float* rstd, float* sums, int64_t rows, int C, int relu, int dtype, void* ws, size_t ws_bytes,
                       iseg_stream_t stream);
int iseg_bn_bwd_apply(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* y, int64_t ldy, const float* mean,
                      const float* rstd, const float* gamma, const float* sums, float inv_n, void* dx, int64_t lddx, int64_t rows,
                      int C, int relu, int dtype, iseg_stream_t stream);
/* iseg_bn_bwd_apply that also books the layer's parameter gradients: dbeta += sums[0:C], dgamma += sums[C:2C] (either may be NULL).  Only
 * when `sums` are this replica's own (no all-reduce in between): under data parallelism the gradients are summed over replicas later. */
int iseg_bn_bwd_apply_acc(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* y, int64_t ldy, const float* mean,
                          const float* rstd, const float* gamma, const float* sums, float inv_n, void* dx, int64_t lddx, float* dgamma,
                          float* dbeta, int64_t rows, int C, int relu, int dtype, iseg_stream_t stream);
/* The fused-ReLU backward pair for a forward that did not keep its output: the ReLU mask is re-derived from x,
 * (x - mean) * rstd * gamma + beta > 0 (the forward's own expression), instead of being read from y. */
int iseg_bn_bwd_reduce_remask(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, float* sums, int64_t rows, int C, int dtype, void* ws, size_t ws_bytes,
                              iseg_stream_t stream);
int iseg_bn_bwd_apply_remask(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, const float* sums, float inv_n, void* dx, int64_t lddx, float* dgamma,
                             float* dbeta, int64_t rows, int C, int dtype, iseg_stream_t stream);
/* One level of the FPN top-down pathway (layers/fpn.py:46-57) in one pass: out = relu((z - mean) * rstd * gamma + beta) + bilinear_up(x),
 * TF2 half-pixel bilinear (utils/common.py resize_image); z, out [N, Ho, Wo, C], x [N, Hi, Wi, C], C % 8 == 0.  mean / rstd come from
 * iseg_bn_stats + iseg_bn_finalize; the backward is iseg_resize_bilinear_bwd + the _remask pair above. */
int iseg_bn_relu_upsample_add(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const void* x,
                              void* out, int N, int Hi, int Wi, int Ho, int Wo, int C, int dtype, iseg_stream_t stream);
int iseg_rsqrt_eps(const float* var, float eps, float* out, int n, iseg_stream_t stream); /* inference: rstd of moving var */

/* ---------------------------------------------------------------------------------------------------------
 * Exchange step of the data-parallel path: distribution/distribution_utils.py:158-169 all_reduce_values -> ReplicaContext.all_reduce(SUM)
 * (SyncBN statistics layers/keras3/bn.py:60-117, gradient sums), over RCCL on xGMI.  One process per GPU: rank 0 creates the 128-byte id
 * and hands it to the other ranks through the host program; every rank then calls iseg_comm_init.  iseg_allreduce_sum works in place and is
 * stream-ordered.  RCCL is resolved at run time (no link dependency); ISEG_ERR_UNSUPPORTED when it cannot be found.
 * (The Python host of this repository keeps using torch.distributed, which drives the same RCCL -- iseg_amd/dist.py.)
 * --------------------------------------------------------------------------------------------------------- */
typedef void* iseg_comm_t;
int iseg_comm_unique_id(void* id128);
int iseg_comm_init(iseg_comm_t* comm, int world_size, int rank, const void* id128);
int iseg_allreduce_sum(iseg_comm_t comm, void* buf, size_t count, int dtype, iseg_stream_t stream);
int iseg_comm_destroy(iseg_comm_t comm);

/* ---------------------------------------------------------------------------------------------------------
 * Layout / elementwise
 * --------------------------------------------------------------------------------------------------------- */
/* Deferred parameter-gradient reductions.  Keras accumulates nothing across ops -- this is a scheduling service of the library: between
 * begin and end, every two-stage reduction of an entry point (iseg_layernorm_bwd, iseg_dwconv2d_bwd_weight, iseg_colsum) whose output lies
 * inside [grad_base, grad_base + grad_bytes) and accumulates into it keeps its partial sums in `arena` (>= 1 MiB, caller-owned) instead of the
 * call's workspace and is queued; iseg_deferred_flush runs one launch over the queue (same fixed summation order as the immediate reduce).
 * The caller flushes before anything reads the gradient buffer (optimizer, clipping, all-reduce).  Single stream, not re-entrant. */
int iseg_deferred_begin(void* grad_base, size_t grad_bytes, void* arena, size_t arena_bytes, iseg_stream_t stream);
int iseg_deferred_flush(iseg_stream_t stream);
int iseg_deferred_end(iseg_stream_t stream);
int iseg_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, iseg_stream_t stream);
/* K-contiguous copies of the bf16 compute kernels for the forward GEMMs (keras Dense / 1x1 Conv2D kernels are [K][N]; with both operands
 * K-contiguous the LDS-DMA GEMM serves the forward pass too).  `count` bf16 matrices inside `src`, transposed into `dst`;
 * table (device, int64 [count][4]) = {src element offset, dst element offset, rows, cols}; max_tiles = max over matrices of
 * ceil(rows/64) * ceil(cols/64).  One launch per optimizer step (iseg_amd/nn.py wt()). */
int iseg_transpose_batched(const void* src, void* dst, const int64_t* table, int count, int max_tiles, iseg_stream_t stream);
/* dst[k][n] = src[k][n]*colscale[n]: layer-scale folded Dense kernel for the dgrad of backbones/convnext.py:54-57 */
int iseg_scale_cols_cast(const float* src, const float* colscale, void* dst, int64_t rows, int cols, int dst_dtype,
                         iseg_stream_t stream);
/* patches of keras.layers.Conv2D(padding="same"/"valid", strides, dilation_rate): col[m][(i*KW+j)*C+c], row stride ldc */
int iseg_im2col(const void* x, int in_dtype, void* col, int out_dtype, int N, int H, int W, int C, int KH, int KW, int sh, int sw,
                int dh, int dw, int pt, int pl, int Ho, int Wo, int64_t ldc, iseg_stream_t stream);
int iseg_col2im(const void* dcol, void* dx, int N, int H, int W, int C, int KH, int KW, int sh, int sw, int dh, int dw, int pt,
                int pl, int Ho, int Wo, int64_t ldc, int dtype, iseg_stream_t stream);
/* out[b][c] (+)= scale * sum_r x[b][r][c]: bias gradients; tf.reduce_mean over H,W (layers/model_builder.py:266) */
size_t iseg_colsum_workspace_bytes(int batch, int64_t rows, int C);
int iseg_colsum(const void* x, int64_t ldx, int64_t batch_stride, int batch, int64_t rows, int C, float* out, float scale,
                int accumulate, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);
/* y[b][r][:] (+)= scale*v[b][:]: ImageLevelBlock broadcast (layers/model_builder.py:268) and the pooling gradient */
int iseg_broadcast_rows(const void* v, int v_dtype, void* y, int64_t ldy, int64_t batch_stride, int batch, int64_t rows, int C,
                        float scale, int accumulate, int dtype, iseg_stream_t stream);
int iseg_axpby(const void* a, const void* b, void* y, float alpha, float beta, int64_t n, int dtype, iseg_stream_t stream);
/* dst0[i] += src[i], dst1[i] += src[n+i] (NULL destinations are skipped): dbeta | dgamma of a normalisation layer's packed backward sums */
int iseg_accumulate_pair(const float* src, int n, float* dst0, float* dst1, iseg_stream_t stream);
/* y = x * s_dev[0] (chain rule with a device-resident scalar; no host read) */
int iseg_scale_dev(const void* x, const float* s_dev, void* y, int64_t n, int dtype, iseg_stream_t stream);
int iseg_rowscale(const void* x, const float* s, void* y, int64_t rows, int C, int64_t rows_per_group, int dtype,
                  iseg_stream_t stream);
/* keras.layers.Dropout: y = x*mask/(1-rate); mask = f(seed, index) so backward = same call on dy.
 * seed_offset (device pointer, may be NULL) is added to `seed` by the kernel: a training step replayed from a HIP graph has frozen launch
 * arguments, so its draw counter lives in device memory (iseg_amd/graphs.py); same meaning in the two drop-path entry points. */
int iseg_dropout(const void* x, void* y, int64_t n, float rate, uint64_t seed, const uint64_t* seed_offset, int dtype, iseg_stream_t stream);
/* utils/drops.py:14-20: s[n] = floor(keep + u_n)/keep */
int iseg_drop_path_mask(float* s, int n, float keep_prob, uint64_t seed, const uint64_t* seed_offset, iseg_stream_t stream);
/* P masks of n samples each in one launch (s [P, n], keep_probs [P] on the device): the drop_path call sites of one training step */
int iseg_drop_path_masks(float* s, const float* keep_probs, int P, int n, uint64_t seed, const uint64_t* seed_offset, iseg_stream_t stream);
int iseg_fill_f32(float* p, float value, int64_t n, iseg_stream_t stream);
/* DCNv2's sampling half (layers/dcn_v2.py:114-229; FaPN's FeatureAlignment, layers/fapn.py:44-80): offset [N,H,W,27] = the offset convolution's output
 * (9 x (dy, dx), then 9 mask logits); col [N,H,W,9,C] = sigmoid(logit_p) * bilinear sample of x (zero-padded by 1) at (h + ky + dy_p, w + kx + dx_p), with the
 * reference's clipping of the position and both corners to [0, H+1] x [0, W+1] and weights from the clipped values.  The layer's output is then the GEMM
 * col [N H W, 9 C] x kernel [9 C, filters] (:230-240).  bwd: dx fp32 (64-bit fixed-point accumulation: deterministic), doffset [N,H,W,27] in storage type. */
int iseg_dcnv2_sample_fwd(const void* x, const void* offset, void* col, int N, int H, int W, int C, int dtype, iseg_stream_t stream);
size_t iseg_dcnv2_sample_bwd_workspace_bytes(int N, int H, int W, int C);
int iseg_dcnv2_sample_bwd(const void* x, const void* offset, const void* dcol, float* dx_f32, void* doffset, int N, int H, int W, int C, int dtype,
                          void* ws, size_t ws_bytes, iseg_stream_t stream);
/* EVA-02 (backbones/eva/*).
 * iseg_qkv_rope: packed attention rows qkv [rows = B * tokens][3 C] = [q | k | v] -> out (out == qkv: in place): q += q_bias, v += v_bias (the fused projection's
 *   bias [q_bias | 0 | v_bias], attention.py:100-112; either may be NULL), then the rotary embedding on q and k of the tokens t >= prefix
 *   (apply_rot_embed_cat, rotar_embedding_cat.py:117-135; the class token keeps its values, attention.py:136-146).  emb fp32 [tokens - prefix][2 head_dim]
 *   = [sin | cos] (RotaryEmbeddingCat.get_embed), NULL = no rotation.  inverse = 1 applies the transposed rotation (the backward pass; pass NULL biases).
 * iseg_glu_fwd / _bwd: out = act(gate) * x on strided [rows, cols] operands (SwiGLU: two Dense outputs, swiglu.py:88-92; GluMlp: the two column halves
 *   of one Dense output, glumlp.py:96-103); act = ISEG_ACT_GELU | ISEG_ACT_SWISH | ISEG_ACT_SIGMOID; bwd: dgate = dout x act'(gate), dx = dout act(gate).
 *   Operands with cols / row strides that are multiples of 8 and 16-byte aligned bases move 16-byte pieces; anything else (EVA02-large: 2730 columns) takes
 *   the one-element-per-lane form. */
int iseg_qkv_rope(const void* qkv, void* out, const float* q_bias, const float* v_bias, const float* emb, int64_t rows, int tokens, int prefix, int C, int head_dim,
                  int inverse, int dtype, iseg_stream_t stream);
int iseg_glu_fwd(const void* gate, int64_t ld_gate, const void* x, int64_t ld_x, void* out, int64_t ld_out, int64_t rows, int cols, int act, int dtype,
                 iseg_stream_t stream);
int iseg_glu_bwd(const void* dout, int64_t ld_dout, const void* gate, int64_t ld_gate, const void* x, int64_t ld_x, void* dgate, int64_t ld_dgate,
                 void* dx, int64_t ld_dx, int64_t rows, int cols, int act, int dtype, iseg_stream_t stream);
/* keras.activations.relu / gelu / sigmoid / swish where no GEMM epilogue is available; bwd: dx = dy*act'(aux), aux = pre-activation */
int iseg_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, iseg_stream_t stream);
int iseg_act_bwd(const void* dy, const void* aux, void* dx, int64_t n, int act, int dtype, iseg_stream_t stream);
/* tf.concat(axis=-1) as slice writes: dst[r][0:cols] = src[r][0:cols] (layers/aspp.py:69) */
int iseg_copy2d(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols, int dtype,
                iseg_stream_t stream);
/* sliding-window inference accumulator (core_inference.py:254-301): dst[r][c] += src[r][c]; y[r][:] = x[r][:]*s[r] */
int iseg_add2d_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t cols, iseg_stream_t stream);
int iseg_scale_rows_f32(const float* x, const float* s, float* y, int64_t rows, int C, iseg_stream_t stream);
/* ConvNeXt layer-scale parameter gradients from Z = g^T dout (see iseg_amd/blocks.py) */
size_t iseg_layerscale_grads_workspace_bytes(int K, int N);
int iseg_layerscale_grads(const float* Z, const float* W2, const float* b2, const float* gamma, const float* S, float* dW2,
                          float* dgamma, float* db2, int K, int N, int accumulate, void* ws, size_t ws_bytes,
                          iseg_stream_t stream);
/* The same from the UNREDUCED split-K slabs of the product Z = g^T dout (iseg_gemm with defer_reduce on an orientation that carries the
 * ones-row: slabs [nslabs][K + 1][N] fp32, row K = column sums of dout): Z and S are formed in slab order while they are read, so the slab-sum
 * launch and the Z tensor disappear.  N % 4 == 0. */
int iseg_layerscale_grads_slabs(const float* slabs, int nslabs, const float* W2, const float* b2, const float* gamma, float* dW2, float* dgamma,
                                float* db2, int K, int N, int accumulate, void* ws, size_t ws_bytes, iseg_stream_t stream);
/* The same plus, in the SAME launch, the slab sum of the pair launch's other product (iseg_gemm_tn_pair: dW1 = y2^T dH with its ones-row db1,
 * backbones/convnext.py:51-57 backward): out0[j] (+)= sum_p partials2[p * n2 + j] for j < n0, out1[j - n0] likewise beyond (out1 may be null) --
 * what iseg_gemm_reduce would have launched (same slab order, same arithmetic).  partials2 must not alias ws; n2, n0 % 4 == 0, 16-byte aligned. */
int iseg_layerscale_grads_slabs_reduce(const float* slabs, int nslabs, const float* W2, const float* b2, const float* gamma, float* dW2,
                                       float* dgamma, float* db2, int K, int N, int accumulate, void* ws, size_t ws_bytes,
                                       const float* partials2, int P2, int64_t n2, float* out0, float* out1, int64_t n0, int accumulate2,
                                       iseg_stream_t stream);
/* The same when the product Z was formed from the ROW-SCALED activation (s g)^T dout and carries no ones-row (slabs [nslabs][K][N]): the column sums
 * S = sum_r rowscale[r / rows_per_group] dout[r][:] that dgamma / db2 need are formed by further workgroups of the same launch from the unscaled
 * bf16 gradient dout [M, ld_dout] and the drop-path factors (utils/drops.py:8-22; rowscale NULL = 1) -- no scaled copy of dout exists.
 * partials2 may be NULL (no second slab job).  N % 8 == 0, N <= 2048. */
int iseg_layerscale_grads_slabs_srow(const float* slabs, int nslabs, const float* W2, const float* b2, const float* gamma, float* dW2,
                                     float* dgamma, float* db2, int K, int N, int accumulate, void* ws, size_t ws_bytes,
                                     const float* partials2, int P2, int64_t n2, float* out0, float* out1, int64_t n0, int accumulate2,
                                     const void* dout, int64_t ld_dout, int64_t M, const float* rowscale, int64_t rows_per_group,
                                     iseg_stream_t stream);
/* ---------------------------------------------------------------------------------------------------------
 * tf.image.resize (half-pixel, no antialias): utils/common.py:107-134 resize_image
 * --------------------------------------------------------------------------------------------------------- */
int iseg_resize_bilinear_fwd(const void* x, int in_dtype, void* y, int out_dtype, int N, int Hi, int Wi, int Ho, int Wo, int C,
                             iseg_stream_t stream);
size_t iseg_resize_bilinear_bwd_workspace_bytes(int N, int Hi, int Wi, int Ho, int Wo, int C);
int iseg_resize_bilinear_bwd(const void* dy, int dy_dtype, void* dx, int dx_dtype, const void* dx_add, int N, int Hi, int Wi,
                             int Ho, int Wo, int C, void* ws, size_t ws_bytes, iseg_stream_t stream);
/* tf.compat.v1.image.resize(x, size, method="bilinear", align_corners=True) (backbones/hrnet.py:303-304, 523-524): source coordinate
 * dst*(in-1)/(out-1), same lerp order; the backward uses iseg_resize_bilinear_bwd_workspace_bytes */
int iseg_resize_bilinear_ac_fwd(const void* x, int in_dtype, void* y, int out_dtype, int N, int Hi, int Wi, int Ho, int Wo, int C,
                                iseg_stream_t stream);
int iseg_resize_bilinear_ac_bwd(const void* dy, int dy_dtype, void* dx, int dx_dtype, const void* dx_add, int N, int Hi, int Wi, int Ho, int Wo,
                                int C, void* ws, size_t ws_bytes, iseg_stream_t stream);
int iseg_resize_nearest_i32(const int32_t* x, int32_t* y, int N, int Hi, int Wi, int Ho, int Wo, int C, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * losses/catecrossentropy_ignore_label.py:44-88 weighted_loss (CategoricalCrossentropy(from_logits), NONE) and
 * metrics/seg_metric_wrapper.py:89-102 + metrics/confusion_matrix.py:65-143.
 * logits [P,C] fp32, labels [P] int32.  loss_px [P] (optional), loss_sum[0] = loss_sum_scale * sum_p loss_p,
 * dlogits = grad_scale * grad_px[p] * w_p * (softmax - onehot) (optional; grad_px NULL = 1).
 * --------------------------------------------------------------------------------------------------------- */
size_t iseg_softmax_ce_workspace_bytes(int64_t P, int C);
int iseg_softmax_ce_ignore(const float* logits, const int32_t* labels, const float* class_w, int64_t P, int C, int ignore_label,
                           float* loss_px, float* loss_sum, float loss_sum_scale, float* dlogits, float grad_scale,
                           const float* grad_px, void* ws, size_t ws_bytes, iseg_stream_t stream);
/* iseg_softmax_ce_ignore that also performs iseg_argmax_confusion on the same logits / labels (cm[y*C + argmax] += 1 for kept,
 * in-range labels; first maximal index): the training step's loss and running-mIoU update in ONE pass over the logits */
int iseg_softmax_ce_confusion(const float* logits, const int32_t* labels, const float* class_w, int64_t P, int C, int ignore_label,
                              float* loss_px, float* loss_sum, float loss_sum_scale, float* dlogits, float grad_scale,
                              const float* grad_px, uint64_t* cm, void* ws, size_t ws_bytes, iseg_stream_t stream);
/* focal variant (use_focal_loss of losses/catecrossentropy_ignore_label.py:27-37 = keras CategoricalFocalCrossentropy, from_logits):
 * p = clip(softmax(logits)[y], 1e-7, 1 - 1e-7);  loss = w * alpha * (1 - p)^gamma * (-log p);  same outputs and masking rules */
int iseg_softmax_focal_ce_ignore(const float* logits, const int32_t* labels, const float* class_w, int64_t P, int C, int ignore_label,
                                 float alpha, float gamma, float* loss_px, float* loss_sum, float loss_sum_scale, float* dlogits,
                                 float grad_scale, const float* grad_px, void* ws, size_t ws_bytes, iseg_stream_t stream);
/* pred_out [P] int32 (optional) = first argmax; cm [C*C] uint64 counts += (label != ignore) at [label][pred] */
int iseg_argmax_confusion(const float* logits, const int32_t* labels, int64_t P, int C, int ignore_label, int32_t* pred_out,
                          unsigned long long* cm, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * optimizers/modern/adamw.py:13-74, optimizers/modern/sgd.py:12-51 over the flat parameter buffer.
 * Every tensor is padded to a multiple of 256 elements; seg_of_block[b] = tensor index of 256-element block b
 * (-1 = padding).  hp (device): [lr, sqrt(1-b2^t)/(1-b1^t), grad_scale, clipvalue(<=0 off)].
 *   AdamW_EXT: g <- NaN->0 (adamw.py:63-74) -> clip -> decoupled decay -> m, v (, v_hat = max(v_hat, v) when vhat != NULL: amsgrad,
 *              adamw.py:54-57) -> w.       SGD_EXT: no NaN scrub; nesterov as sgd.py:46-49.
 *   Clipping = Keras' base optimizer (_clip_gradients), which the reference drives through get_optimizer(clipnorm, clipvalue)
 *   (core_optimizer.py:170-183): clipnorm is PER VARIABLE (tf.clip_by_norm), global_clipnorm over all variables
 *   (tf.clip_by_global_norm), clipvalue per element; at most one of them is active.  The squared norms come from iseg_grad_sqnorm
 *   (fixed-order sums: 256-element blocks -> variables -> total; seg_first_block [nseg + 1] = first block of each variable;
 *   block_ws [nblocks] floats of scratch; seg_l2 / w as in the SGD step or NULL).
 * --------------------------------------------------------------------------------------------------------- */
int iseg_grad_sqnorm(const float* g, const float* w, const int32_t* seg_of_block, const int32_t* seg_first_block, const float* seg_l2,
                     const float* hp, int scrub_nan_grads, float* block_ws, float* seg_sq, float* global_sq, int64_t nblocks, int nseg,
                     iseg_stream_t stream);
int iseg_adamw_step(float* w, const float* g, float* m, float* v, float* vhat, void* w_bf16, const int32_t* seg_of_block,
                    const float* seg_lr_mult, const float* seg_wd, const float* hp, float beta1, float beta2, float eps,
                    const float* seg_sq, float clipnorm, const float* global_sq, float global_clipnorm, int64_t nblocks,
                    iseg_stream_t stream);
int iseg_sgd_momentum_step(float* w, const float* g, float* m, void* w_bf16, const int32_t* seg_of_block, const float* seg_lr_mult,
                           const float* seg_l2, const float* hp, float momentum, int nesterov, const float* seg_sq, float clipnorm,
                           const float* global_sq, float global_clipnorm, int64_t nblocks, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * utils/op_utils.py:43-60 replace_nan_or_inf (layers/fpn.py:52): NaN -> nan_value, then clip to the [min, max] of the
 * tensor with +-inf counted as 0.  ws: 8 bytes.  Gradient passes where x is finite.
 * --------------------------------------------------------------------------------------------------------- */
int iseg_replace_nan_or_inf(const void* x, void* y, int64_t n, float nan_value, int dtype, void* ws, size_t ws_bytes,
                            iseg_stream_t stream);
int iseg_replace_nan_or_inf_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * layers/groupnorm.py:148-207 GroupNormalization (axis=-1): x [N,HW,C], moments over (HW, C/G) per (n, g);
 * mean, rstd [N*G].  layers/rmsnorm.py:22-29 RMSNormalization: y = x * rsqrt(mean_c(x^2)+eps) * (1+scale), rstd [rows].
 * --------------------------------------------------------------------------------------------------------- */
int iseg_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int N, int HW,
                       int C, int G, float eps, int dtype, iseg_stream_t stream);
size_t iseg_groupnorm_bwd_workspace_bytes(int N, int C);
int iseg_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                       float* dgamma, float* dbeta, int accumulate_param_grads, int N, int HW, int C, int G, int dtype, void* ws,
                       size_t ws_bytes, iseg_stream_t stream);
int iseg_rmsnorm_fwd(const void* x, const float* scale, void* y, float* rstd, int64_t rows, int C, float eps, int dtype,
                     iseg_stream_t stream);
size_t iseg_rmsnorm_bwd_workspace_bytes(int64_t rows, int C);
int iseg_rmsnorm_bwd(const void* dy, const void* x, const float* scale, const float* rstd, void* dx, float* dscale,
                     int accumulate_param_grads, int64_t rows, int C, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Global Response Normalization of ConvNeXt V2: backbones/convnext_v2.py:17-60 GlobalResponseNormlizationLayer.call,
 *   gx[n,c] = sqrt(sum_hw x^2 + eps);  nx = gx / (mean_c gx + eps);  y = gamma*(x*nx) + beta + x      (fp32 arithmetic, output in x's dtype)
 * x, y [N,HW,C] bf16 or fp32 (C % 8 == 0); gamma, beta [C] fp32; nx, gx [N,C] fp32 are written by the forward and read by the backward.
 * The backward returns dx and (+)= dgamma, dbeta; `mul` (NULL, or a tensor shaped like x) multiplies dx elementwise -- the saved derivative of
 * the GELU in front of the layer (Block.call :91-92), so that chain-rule step needs no pass of its own.  One workspace size serves both
 * directions.  All reductions run in a fixed order.
 * --------------------------------------------------------------------------------------------------------- */
size_t iseg_grn_workspace_bytes(int64_t N, int64_t HW, int C);
int iseg_grn_fwd(const void* x, const float* gamma, const float* beta, void* y, float* nx, float* gx, int64_t N, int64_t HW, int C,
                 float eps, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);
int iseg_grn_bwd(const void* dy, const void* x, const float* gamma, const float* nx, const float* gx, const void* mul, void* dx,
                 float* dgamma, float* dbeta, int accumulate_param_grads, int64_t N, int64_t HW, int C, float eps, int dtype, void* ws, size_t ws_bytes,
                 iseg_stream_t stream);
/* The normalisation folded into the Dense that follows it (Block.call :92-93, x = grn(x); x = pwconv2(x)): with a_n = gamma*nx_n + 1 per sample,
 *   grn(g) W + b = g (diag(a_n) W) + (b + beta W),   dW = sum_n diag(a_n) (g_n^T dbr_n) + beta (x) colsum(dbr),
 * and the statistics of the GRN backward follow from the per-sample products g_n^T dbr_n as well -- the passes that would write grn(g) and
 * read it back disappear.  iseg_grn_fwd with y == NULL only computes nx, gx.
 *   iseg_grn_fold_weights: out[n][o][k] = wt[o][k] * a_n[k]    (wt: the K-contiguous bf16 kernel copy [Cout][C4]; out feeds iseg_gemm's B per row group)
 *   iseg_grn_fold_bias:    out[c] = b[c] + sum_k beta[k] W[k][c]                      (W: the fp32 kernel [C4][Cout])
 *   iseg_grn_fold_wgrad:   slabs [N*slabs_per_sample][C4][Cout] fp32 = g^T dbr of consecutive equal row chunks (a batched iseg_gemm), S = colsum(dbr):
 *                          dW (+)= sum_n a_n (.) G_n + beta (x) S;  dstats [N][2*C4] = (sum_hw dz*g | sum dz in row 0, zeros below)
 *   iseg_grn_bwd_folded:   iseg_grn_bwd with those statistics given (dstats is overwritten): no pass over dy for them */
int iseg_grn_fold_weights(const void* wt, const float* gamma, const float* nx, void* out, int64_t N, int Cout, int C4, iseg_stream_t stream);
int iseg_grn_fold_bias(const float* W, const float* beta, const float* b, float* out, int C4, int Cout, iseg_stream_t stream);
int iseg_grn_fold_wgrad(const float* slabs, int slabs_per_sample, const float* W, const float* gamma, const float* beta, const float* nx,
                        const float* S, float* dW, float* dstats, int accumulate, int64_t N, int C4, int Cout, iseg_stream_t stream);
int iseg_grn_bwd_folded(const void* dy, const void* x, const float* gamma, const float* nx, const float* gx, const void* mul, float* dstats,
                        void* dx, float* dgamma, float* dbeta, int accumulate_param_grads, int64_t N, int64_t HW, int C, float eps, int dtype,
                        void* ws, size_t ws_bytes, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * keras MaxPooling2D(3, strides=2, "same") backbones/resnet_common.py:215-217 and tf.nn.avg_pool2d(..., "SAME")
 * backbones/resnet_blocks.py:182-186.  mode 0 = max (gradient to the first maximal cell), 1 = average over valid cells.
 * --------------------------------------------------------------------------------------------------------- */
int iseg_pool2d_fwd(const void* x, void* y, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int pad_t, int pad_l,
                    int Ho, int Wo, int mode, int dtype, iseg_stream_t stream);
/* relu(a + b), n % 8 == 0: residual join of backbones/resnet_blocks.py:106-107,202-203 */
int iseg_add_relu(const void* a, const void* b, void* y, int64_t n, int dtype, iseg_stream_t stream);
/* max mode with C % 8 == 0 runs two vector passes through a one-byte-per-output-element winner table (workspace); everything else
   takes the one-pass gather kernel and needs no workspace (ws may be NULL when iseg_pool2d_bwd_workspace_bytes returns 0) */
size_t iseg_pool2d_bwd_workspace_bytes(int N, int Ho, int Wo, int C, int mode);
int iseg_pool2d_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int pad_t,
                    int pad_l, int Ho, int Wo, int mode, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Attention pieces around the strided-batch GEMMs (iseg_gemm with batch > 1):
 *   softmax over keys of scores [problems, Tq, ld] (cols valid, pad columns written as 0) with the additive terms of
 *   backbones/swin.py:131-158 -- bias [heads,Tq,cols] fp32 (problem z uses head z % heads) and shift mask [windows,Tq,cols]
 *   fp32 (window (z / heads) % windows) -- and the optional probability clip of layers/multihead_self_attention.py:138
 *   (clip_hi > clip_lo enables it).  Backward: dS = P * (g - sum_j g_j P_j), g = dP where the clip passed.
 *   Keras MultiHeadAttention (backbones/vit.py:142-147) uses the same kernels without bias / mask / clip.
 * --------------------------------------------------------------------------------------------------------- */
int iseg_softmax_rows_fwd(const void* scores, void* probs, int64_t problems, int Tq, int cols, int ld, const float* bias, int heads,
                          const float* mask, int windows, float clip_lo, float clip_hi, int dtype, iseg_stream_t stream);
int iseg_softmax_rows_bwd(const void* probs, const void* dprobs, void* dscores, int64_t rows, int cols, int ld, float clip_lo,
                          float clip_hi, int dtype, iseg_stream_t stream);
/* tf.clip_by_value and its gradient (passes where lo <= x <= hi) */
int iseg_clip_fwd(const void* x, void* y, int64_t n, float lo, float hi, int dtype, iseg_stream_t stream);
int iseg_clip_bwd(const void* x, const void* dy, void* dx, int64_t n, float lo, float hi, int dtype, iseg_stream_t stream);
/* y[r,:] = idx[r] >= 0 ? x[idx[r],:] : 0 -- tf.pad / tf.roll / window_partition / window_reverse / crop of
 * backbones/swin.py:46-64,258-288 and PatchMerging's 2x2 space-to-depth (:316-327) as one row permutation each */
int iseg_gather_rows(const void* x, const int32_t* idx, void* y, int64_t rows_in, int64_t rows_out, int C, int dtype,
                     iseg_stream_t stream);
/* y[r,:] = residual[r,:] + scale[g] * (idx[r] >= 0 ? x[idx[r],:] : 0), g = (scale_by_source_row ? idx[r] : r) / rows_per_group; residual
 * and scale optional, C % 8 == 0 -- window_reverse + roll back + crop + drop path + skip connection of a Swin block
 * (backbones/swin.py:264-279, utils/drops.py:8-22) in one pass, and its gradient towards the window rows (inverse table,
 * scale_by_source_row = 1) */
int iseg_gather_rows_fma(const void* x, const int32_t* idx, const float* scale, int64_t rows_per_group, int scale_by_source_row,
                         const void* residual, void* y, int64_t rows_out, int C, int dtype, iseg_stream_t stream);
/* backbones/swin.py:134-142: bias[h,i,j] = table[index[i,j], h]; gradient dtable[k,h] (+)= sum_{index[i,j]==k} dbias[h,i,j]
 * (dbias rows have stride ld) */
/* out[c] (+)= sum_r x[r*ldx + c] for a short, very wide matrix (the [windows, heads*T*ld] score gradients -> bias gradient) */
size_t iseg_colsum_wide_workspace_bytes(int64_t rows, int64_t cols);
int iseg_colsum_wide(const void* x, int64_t ldx, int64_t rows, int64_t cols, float* out, int accumulate, int dtype, void* ws,
                     size_t ws_bytes, iseg_stream_t stream);
int iseg_relpos_bias_gather(const float* table, const int32_t* index, float* bias, int heads, int TT, iseg_stream_t stream);
int iseg_relpos_bias_scatter_grad(const float* dbias, int ld, const int32_t* index, float* dtable, int entries, int heads, int T,
                                  int accumulate, iseg_stream_t stream);
/* the same gradient when `index` is the canonical square-window table of backbones/swin.py:93-104 (window side ws, T = ws*ws,
 * (2ws-1)^2 entries): displacement-enumerating kernel, no table scan */
int iseg_relpos_bias_scatter_grad_window(const float* dbias, int ld, float* dtable, int ws, int heads, int accumulate,
                                         iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * layers/dcn_v3/op.py:16-109 dcnv3_op + utils.py:14-209 (reference points, dilation grids, bilinear sampler), restated
 * exactly (see csrc/dcnv3.hip for the formulae and the [y,x]-vs-[x,y] quirk).  x [N,H,W,G*Cg] (unpadded; `pad` zero ring is
 * implicit), offset [N,Ho,Wo,G*kh*kw*2], mask [N,Ho,Wo,G*kh*kw] (already soft-maxed), y [N,Ho,Wo,G*Cg].
 * Backward: dx_f32 [N,H,W,G*Cg] fp32 (written in full, no need to clear it); doffset / dmask in the storage dtype.  Group widths 8 and
 * 16 (InternImage: 16) take the deterministic path -- int64 fixed-point LDS windows per (output tile, group), then an ordered gather --
 * and need iseg_dcnv3_bwd_workspace_bytes(...) of scratch (0 = the fallback path, which scatters with fp32 atomics, is taken).
 * --------------------------------------------------------------------------------------------------------- */
int iseg_dcnv3_fwd(const void* x, const void* offset, const void* mask, void* y, int N, int H, int W, int G, int Cg, int kh, int kw,
                   int stride, int dil, int pad, float offset_scale, int dtype, iseg_stream_t stream);
size_t iseg_dcnv3_bwd_workspace_bytes(int N, int H, int W, int G, int Cg, int kh, int kw, int stride, int dil, int pad, float offset_scale);
int iseg_dcnv3_bwd(const void* x, const void* offset, const void* mask, const void* dy, float* dx_f32, void* doffset, void* dmask,
                   int N, int H, int W, int G, int Cg, int kh, int kw, int stride, int dil, int pad, float offset_scale, int dtype,
                   void* ws, size_t ws_bytes, iseg_stream_t stream);
/* The same two entry points with the offsets and the mask (and their gradients) as column ranges of ONE [pixels][ld] matrix -- the output of the
 * layer's offset and mask projections (layers/dcn_v3/dcn_v3.py:116-123, two Dense layers on the same input) run as a single product:
 * ld_off / ld_mask = elements between the runs of consecutive output pixels (0 = dense: 2 G P / G P; ld_off even).
 * iseg_dcn_mask_softmax_fwd / _bwd: the softmax over each (pixel, group)'s P mask logits (dcn_v3.py:121-123) in place on columns
 * [col0, col0 + G P) of such a matrix; the backward also clears the columns behind col0 + G P of the gradient matrix (padding up to the
 * product's width), P <= 9.  iseg_split_cols_accumulate: dst0 [rows][n0] += src[:, :n0], dst1 [rows][n1] += src[:, n0:n0+n1] (fp32) -- the
 * joint weight / bias gradient handed back to the two layers. */
int iseg_dcnv3_fwd_ld(const void* x, const void* offset, const void* mask, int64_t ld_off, int64_t ld_mask, void* y, int N, int H, int W, int G,
                      int Cg, int kh, int kw, int stride, int dil, int pad, float offset_scale, int dtype, iseg_stream_t stream);
/* _bwd_ld further takes the storage type of dx (ISEG_F32, or ISEG_BF16: the input gradient is written in the activations' type, no fp32 image
 * and cast pass) and, optionally, a caller-kept side buffer of iseg_dcnv3_bwd_side_bytes(...) bytes that is ALL ZERO on entry and is left all zero
 * (the per-call zeroing of 8 bytes per input element goes away; null = the side buffer lives in `ws` and is zeroed per call). */
size_t iseg_dcnv3_bwd_side_bytes(int N, int H, int W, int G, int Cg, int kh, int kw, int stride, int dil, int pad, float offset_scale);
int iseg_dcnv3_bwd_ld(const void* x, const void* offset, const void* mask, int64_t ld_off, int64_t ld_mask, const void* dy, void* dx, int dx_dtype,
                      void* doffset, void* dmask, int N, int H, int W, int G, int Cg, int kh, int kw, int stride, int dil, int pad,
                      float offset_scale, int dtype, void* ws, size_t ws_bytes, void* side_keep, size_t side_keep_bytes, iseg_stream_t stream);
int iseg_dcn_mask_softmax_fwd(void* om, int64_t pixels, int G, int P, int64_t ld, int col0, int dtype, iseg_stream_t stream);
int iseg_dcn_mask_softmax_bwd(const void* om, void* dom, int64_t pixels, int G, int P, int64_t ld, int col0, int dtype, iseg_stream_t stream);
int iseg_split_cols_accumulate(const float* src, int64_t rows, int64_t ld, float* dst0, int n0, float* dst1, int n1, iseg_stream_t stream);
/* Centre-feature scale of the DCNv3 layer (layers/dcn_v3/dcn_v3.py:138-146; intern_image_huge): out = x (1 - s) + x_proj s with
 * s [pixels, G] (the un-squashed output of center_feature_scale_proj) broadcast over the Cg channels of its group; backward: dx, dx_proj and
 * ds [pixels, G] = sum_c dout (x_proj - x).  Cg in {8, 16}. */
int iseg_dcn_center_blend_fwd(const void* x, const void* x_proj, const void* scale, void* out, int64_t pixels, int G, int Cg, int dtype,
                              iseg_stream_t stream);
int iseg_dcn_center_blend_bwd(const void* dout, const void* x, const void* x_proj, const void* scale, void* dx, void* dx_proj, void* dscale,
                              int64_t pixels, int G, int Cg, int dtype, iseg_stream_t stream);

/* out[c] (+)= sum_r a[r][c]*b[r][c]: gradient of the per-channel layer scale x * gamma (backbones/intern_image/
 * intern_image_layer.py:128,136,160,168) */
int iseg_scale_cols(const void* x, const float* colscale, void* y, int64_t rows, int C, int dtype, iseg_stream_t stream);
size_t iseg_mul_colsum_workspace_bytes(int64_t rows, int C);
int iseg_mul_colsum(const void* a, const void* b, int64_t rows, int C, float* out, int accumulate, int dtype, void* ws,
                    size_t ws_bytes, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fused window attention (backbones/swin.py:117-167 WindowAttention.call between the qkv and proj Dense layers):
 *   out[b,:,h] = softmax(scale * q k^T + bias[h] + mask[b % mask_windows]) v    for qkv [windows, T, 3*heads*32] bf16
 * bias [heads,T,T] fp32 (iseg_relpos_bias_gather) and mask [mask_windows,T,T] fp32 (or NULL) are first folded into one padded
 * additive table (iseg_window_attention_table).  One wavefront per (window, head), MFMA
 * for all five products, nothing of size T x T in HBM.  Supported: bf16, head_dim 32, T <= 64 (iseg_window_attention_supported).
 * Backward recomputes the probabilities; dbias [heads,T,T] fp32 is overwritten (sum over windows, fixed order).
 * --------------------------------------------------------------------------------------------------------- */
int iseg_window_attention_supported(int T, int head_dim, int dtype);
/* table[w][h][64][64] fp32 = bias[h] + mask[w] inside T x T, -FLT_MAX outside (w < mask_windows; one window when mask is NULL) */
int iseg_window_attention_table(const float* bias, const float* mask, float* table, int T, int heads, int mask_windows,
                                iseg_stream_t stream);
int iseg_window_attention_fwd(const void* qkv, const float* table, void* out, int64_t windows, int T, int heads, int table_windows,
                              float scale, int dtype, iseg_stream_t stream);
size_t iseg_window_attention_bwd_workspace_bytes(int64_t windows, int T, int heads);
int iseg_window_attention_bwd(const void* qkv, const float* table, const void* dout, void* dqkv, float* dbias, int64_t windows, int T,
                              int heads, int table_windows, float scale, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Forward-only global self-attention for inference (keras MultiHeadAttention of backbones/vit.py:142-147,166 under
 * core_inference.py's sliding windows): out[b,:,h] = softmax(scale * q k^T) v for packed qkv [batch, T, 3*heads*64] bf16.
 * Online softmax over key tiles of 64: no T x T tensor in HBM.  Supported: bf16, head_dim 64, any T.
 * --------------------------------------------------------------------------------------------------------- */
int iseg_attention_fwd_supported(int head_dim, int dtype);
int iseg_attention_fwd(const void* qkv, void* out, int64_t batch, int T, int heads, int head_dim, float scale, int dtype,
                       iseg_stream_t stream);
/* Training pair of the same kernel family (same support set).  The forward additionally returns, per (sample, head, token) and
 * padded to a multiple of 64 tokens (iseg_attention_lse_elems floats), log2 of the softmax denominator in the kernel's exp2 domain;
 * the backward recomputes the probabilities from qkv and that vector, tile by tile (dQ kernel over key tiles, dK/dV kernel over
 * query tiles: no atomics, fixed summation order).  dqkv has the layout of qkv and is overwritten; workspace = one float per
 * lse element. */
size_t iseg_attention_lse_elems(int64_t batch, int T, int heads);
int iseg_attention_fwd_train(const void* qkv, void* out, float* lse2, int64_t batch, int T, int heads, int head_dim, float scale,
                             int dtype, iseg_stream_t stream);
size_t iseg_attention_bwd_workspace_bytes(int64_t batch, int T, int heads);
int iseg_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse2, void* dqkv, int64_t batch, int T,
                       int heads, int head_dim, float scale, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * On-device input pipeline: data_process/pipeline.py:85-170 (StandardAugmentationsPipeline, training branch) + data_process/input_norm.py:7-80
 * as one gather per output pixel -- random scale (bilinear image / nearest label, utils.py:303-370), bottom / right pad with the mean
 * pixel / ignore label (augments/pad_augment.py), random crop, random flip, random erasing with noise (augments/random_erasing_augment.py)
 * and out = v * norm_scale[c] + norm_shift[c].  images [B, Hs, Ws, 3] float32 (dtype 0) or uint8 (dtype 2), zero-padded to a common size;
 * labels [B, Hs, Ws] int32 or NULL; params [B, iseg_augment_params_ints()] int32 on the device, per sample:
 * H, W (valid source size), newH, newW (after the scale), off_y, off_x (crop offset in the scaled + padded image), flip, n_erase (<= 5),
 * then n_erase x (y, x, h, w) in output coordinates.  The host draws them; the device only draws the erase noise (seed).
 * Optional photometric table fparams [B, iseg_augment_params_floats()] float32 (NULL = none; round 3), per sample: brightness delta
 * (augments/random_brightness_augment.py: + delta, clip [0, 256]; 0 = not executed), contrast factor (tf.image.adjust_contrast about the
 * channel means in slots 2..4, which iseg_augment_channel_means fills from the scaled + brightness-adjusted image; 1 = not executed),
 * saturation factor (1) and hue delta (0) (augments/random_photo_metric_distortions.py: contrast -> saturation -> hue -> clip), and the
 * stddev of RandomNoisyEvalAugment's N(0, sigma) noise (evaluation pipeline: added after the padding, clip [0, 256]).  The photometric
 * steps sit between the random scale and the padding (pipeline.py:129-134): pad and erased pixels are not touched by them.
 * --------------------------------------------------------------------------------------------------------- */
int iseg_augment_params_ints(void);
int iseg_augment_params_floats(void);
size_t iseg_augment_means_workspace_bytes(int B);
int iseg_augment_channel_means(const void* images, int image_dtype, const int32_t* params, float* fparams, int B, int Hs, int Ws, void* ws,
                               size_t ws_bytes, iseg_stream_t stream);
int iseg_augment_crop_batch(const void* images, int image_dtype, const int32_t* labels, const int32_t* params, const float* fparams,
                            const float* mean_pixel, const float* norm_scale, const float* norm_shift, int ignore_label, float* out_images,
                            int32_t* out_labels, int B, int Hs, int Ws, int crop_h, int crop_w, uint64_t seed, iseg_stream_t stream);
int iseg_normalize_image(const float* x, float* y, int64_t pixels, const float* norm_scale, const float* norm_shift, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fused logits tail of the training step: tf.image.resize(bilinear) of the low-resolution logits z [N,Hi,Wi,C] to the label size
 * (layers/core_model_ext.py:199-256) + the ignore-label cross-entropy mean and its gradient w.r.t. z
 * (losses/catecrossentropy_ignore_label.py:44-88) + the argmax confusion matrix (metrics/seg_metric_wrapper.py:89-102), in one pass that
 * never writes the [N,Ho,Wo,C] fp32 logits or their gradient.  Same arithmetic as iseg_resize_bilinear_fwd followed by
 * iseg_softmax_ce_confusion followed by iseg_resize_bilinear_bwd (TF's lerp order, first-max argmax, zero one-hot row for labels
 * outside [0,C)), deterministic.  Needs an integer factor (rows even, columns a power of two <= 64) and C <= 32
 * (iseg_upsample_ce_supported); other shapes take the three separate calls.
 *   loss_sum[0] = loss_sum_scale * sum_p w_p (logsumexp(l_p) - l_p[y_p]);  dz (same dtype as z, or NULL) = grad_scale * d(sum)/dz;
 *   cm [C,C] uint64 (or NULL) += counts.
 * --------------------------------------------------------------------------------------------------------- */
int iseg_upsample_ce_supported(int Hi, int Wi, int Ho, int Wo, int C);
size_t iseg_upsample_ce_workspace_bytes(int N, int Hi, int Wi, int Ho, int Wo, int C);
int iseg_upsample_ce(const void* z, int dtype, const int32_t* labels, const float* class_w, int N, int Hi, int Wi, int Ho, int Wo, int C,
                     int ignore_label, float* loss_sum, float loss_sum_scale, void* dz, float grad_scale, uint64_t* cm, void* ws,
                     size_t ws_bytes, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the matrix cores: keras.layers.Conv2D(filters, k, strides, padding="same", dilation_rate,
 * groups, use_bias) as built by layers/model_builder.py:54-64 (ConvNormAct.conv), layers/aspp.py:41-52 (dilated 3x3 branches),
 * backbones/resnet_blocks.py:175-205, backbones/convnext.py:72-75,255-257 (2x2/s2 or dilated downsample), layers/simpledecoder.py:21-36.
 * x [N,H,W,Cin], w [KH,KW,Cin/groups,Cout] (Keras layout, bf16 shadow), y [N,Ho,Wo,Cout]; pt / pl = TF "same" top / left padding
 * (the halo is zero chunks of the gathered operand -- no padded copy, no [pixels, KH*KW*Cin] column buffer, no col2im scatter).
 * bf16 storage, channels per group multiples of 8 (iseg_conv2d_igemm_supported); fp32 parity runs keep iseg_im2col + iseg_gemm.
 *   fwd         y = conv(x, w) (+ bias)
 *   bwd_data    dx = conv^T(dy, w)          gather form, deterministic
 *   bwd_weight  dw (+)= x^T * dy per tap    fp32 [KH,KW,Cin/groups,Cout], fixed-order split-K slabs
 * Workspace (split-K slabs): iseg_conv2d_igemm_workspace_bytes(geom, pass) with pass 0 fwd, 1 bwd_data, 2 bwd_weight.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct iseg_conv_geom {
    int N, H, W, Cin, Cout, KH, KW, sh, sw, dh, dw, pt, pl, Ho, Wo, groups;
} iseg_conv_geom;
int iseg_conv2d_igemm_supported(const iseg_conv_geom* geom_h, int dtype);
size_t iseg_conv2d_igemm_workspace_bytes(const iseg_conv_geom* geom_h, int pass);
int iseg_conv2d_igemm_fwd(const void* x, const void* w, const float* bias, void* y, const iseg_conv_geom* geom_h, int dtype, void* ws,
                          size_t ws_bytes, iseg_stream_t stream);
/* The forward pass on the LDS-DMA pipeline (csrc/conv_igemm_dma.h): `wt` is the K-contiguous copy [Cout][KH*KW*Cin] of the Keras kernel (what
 * iseg_transpose_batched keeps per weight update); one group, Cin % 64 == 0, Cout % 8 == 0.  iseg_conv2d_igemm_bwd_data takes the same pipeline by
 * itself for stride-1 problems with Cout % 64 == 0 (the Keras kernel already is K-contiguous for that product). */
int iseg_conv2d_igemm_fwd_kt_supported(const iseg_conv_geom* g, int dtype);
int iseg_conv2d_igemm_fwd_kt(const void* x, const void* wt, const float* bias, void* y, const iseg_conv_geom* g, int dtype, void* ws,
                             size_t ws_bytes, iseg_stream_t stream);
int iseg_conv2d_igemm_bwd_data(const void* dy, const void* w, void* dx, const iseg_conv_geom* geom_h, int dtype, void* ws, size_t ws_bytes,
                               iseg_stream_t stream);
int iseg_conv2d_igemm_bwd_weight(const void* x, const void* dy, float* dw, int accumulate, const iseg_conv_geom* geom_h, int dtype, void* ws,
                                 size_t ws_bytes, iseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fused ConvNeXt MLP (backbones/convnext.py:51-63 Block.call: pwconv1 -> act (exact GELU) -> pwconv2 -> gamma -> drop_path ->
 * + input) for the wide stages, bf16 storage, C = 96 / 192 (iseg_convnext_mlp_supported):
 *     out[m][:] = residual[m][:] + rowscale[m / rows_per_group] * gamma[:] * (gelu(y2[m][:] @ W1 + b1) @ W2 + b2)
 * y2 = LayerNorm output [M, C].  The [M, 4C] hidden tensor never reaches HBM in the forward pass and is recomputed by the backward
 * chain.  The kernels stream the weights as bf16 *tiled* images (the byte order of their LDS ring stages):
 *   iseg_convnext_mlp_prep      fp32 Keras kernels W1 [C, 4C], W2 [4C, C] (+ layer scale gamma [C] or NULL) -> fw_tiled
 *                               (iseg_convnext_mlp_tiled_bytes(C, 0) bytes) and, when bw_tiled != NULL, bw_tiled (.. (C, 1) bytes; it
 *                               holds W1, W2 * gamma and W1 again in the orders the backward products read them); once per step
 *   iseg_convnext_mlp_fwd       the formula above; gamma / rowscale (per-sample drop-path factor) may be NULL
 *   iseg_convnext_mlp_bwd       dbr [M, C] = rowscale * d(out)  ->  g [M, 4C] = gelu(h), dh [M, 4C] = (dbr @ (W2 gamma)^T) * gelu'(h)
 *                               (the operands of the two weight-gradient GEMMs) and dy2 [M, C] = dh @ W1^T
 * --------------------------------------------------------------------------------------------------------- */
int iseg_convnext_mlp_supported(int C, int dtype);
size_t iseg_convnext_mlp_tiled_bytes(int C, int backward);
int iseg_convnext_mlp_prep(const float* W1, const float* W2, const float* gamma, void* fw_tiled, void* bw_tiled, int C,
                           iseg_stream_t stream);
/* all per-weight-update derivations of a model's ConvNeXt blocks in one launch.  table (device): per entry 8 x int64
 * {kind, p1, p2, p3, p4, p5, n, rows}: kind 1 = iseg_convnext_mlp_prep(W1 = p1, W2 = p2, gamma = p3 | 0, fw_tiled = p4, bw_tiled = p5 | 0, C = n);
 * kind 0 = iseg_scale_cols_cast(src = p1, colscale = p3, dst(bf16) = p4, rows, cols = n).  max_elements sizes the grid. */
int iseg_convnext_weight_prep_batched(const int64_t* table, int entries, int64_t max_elements, iseg_stream_t stream);
int iseg_convnext_mlp_fwd(const void* y2, const void* fw_tiled, const float* b1, const float* b2, const float* gamma,
                          const float* rowscale, int64_t rows_per_group, const void* residual, void* out, int64_t M, int C, int dtype,
                          iseg_stream_t stream);
int iseg_convnext_mlp_bwd(const void* y2, const void* dbr, const void* bw_tiled, const float* b1, void* g, void* dh, void* dy2,
                          int64_t M, int C, int dtype, iseg_stream_t stream);
/* The same block's backward pass with NO [M, 4C] tensor in HBM (round 3; the training path uses this pair):
 *   iseg_convnext_mlp_bwd_data  dy2 [M, C] only; dbr = rowscale[m / rows_per_group] * dout is formed while the rows are loaded
 *                               (rowscale may be NULL); mean != NULL: `y` is the LayerNorm input y1, normalised on load like in _wgrad
 *   iseg_convnext_mlp_wgrad     every parameter gradient of backbones/convnext.py:51-57 (pwconv1, pwconv2, gamma), ACCUMULATED into
 *                               dW1 [C, 4C], db1 [4C], dW2 [4C, C], db2 [C], dgamma [C]: workgroups own 128 hidden units and a chunk of
 *                               rows, recompute gelu(h) / dh for them and contract over the rows on the matrix cores; partial sums per
 *                               row chunk in `ws` (iseg_convnext_mlp_wgrad_workspace_bytes), summed in chunk order by a second launch
 *                               (deterministic).  With gamma: dW2 += (g^T dbr) gamma, dgamma += sum_k W2 o (g^T dbr) + b2 S,
 *                               db2 += gamma S with S = column sums of dbr; gamma == NULL: dW2 += g^T dbr, db2 += S.  W2 / b2 / gamma
 *                               are the fp32 masters.  mean != NULL: `y` is the LayerNorm INPUT y1 and y2 = (y1 - mean) rstd ln_gamma +
 *                               ln_beta is formed while the rows are staged.  rowscale needs rows_per_group % 64 == 0. */
int iseg_convnext_mlp_bwd_data(const void* y, const float* mean, const float* rstd, const float* ln_gamma, const float* ln_beta,
                               const void* dout, const float* rowscale, int64_t rows_per_group, const void* bw_tiled, const float* b1,
                               void* dy2, int64_t M, int C, int dtype, iseg_stream_t stream);

/* The same chain carried through the LayerNorm in front of the MLP as well (LayerNorm on the row load, backbones/convnext.py:26-27,52): y1 is the
 * LayerNorm INPUT, dy1 receives its gradient -- rstd (t - mean_c(t) - xhat mean_c(t xhat)), t = dy2 gamma -- and the column sums
 * dgamma += sum_rows dy2 xhat, dbeta += sum_rows dy2 are added to dln_gamma / dln_beta (per-workgroup partial rows in ws, fixed-order sum;
 * deferred when a reduction queue is open).  Replaces iseg_convnext_mlp_bwd_data + iseg_layernorm_bwd for the fused stages. */
size_t iseg_convnext_mlp_bwd_data_ln_workspace_bytes(int64_t M, int C);
int iseg_convnext_mlp_bwd_data_ln(const void* y1, const float* mean, const float* rstd, const float* ln_gamma, const float* ln_beta,
                                  const void* dout, const float* rowscale, int64_t rows_per_group, const void* bw_tiled, const float* b1,
                                  void* dy1, float* dln_gamma, float* dln_beta, int64_t M, int C, int dtype, void* ws, size_t ws_bytes,
                                  iseg_stream_t stream);
/* the forward kernel with keras.layers.LayerNormalization(epsilon) (backbones/convnext.py:27,49) folded into its row load: y1 = the
 * depthwise convolution's output; mean / rstd [M] are written for the backward kernels; y2 never exists in HBM */
int iseg_convnext_mlp_fwd_ln(const void* y1, const float* ln_gamma, const float* ln_beta, float eps, float* mean, float* rstd,
                             const void* fw_tiled, const float* b1, const float* b2, const float* gamma, const float* rowscale,
                             int64_t rows_per_group, const void* residual, void* out, int64_t M, int C, int dtype, iseg_stream_t stream);
size_t iseg_convnext_mlp_wgrad_workspace_bytes(int64_t M, int C);
int iseg_convnext_mlp_wgrad(const void* y, const float* mean, const float* rstd, const float* ln_gamma, const float* ln_beta,
                            const void* dout, const float* rowscale, int64_t rows_per_group, const void* bw_tiled, const float* b1,
                            const float* W2, const float* b2, const float* gamma, float* dW1, float* db1, float* dW2, float* db2,
                            float* dgamma, int64_t M, int C, int dtype, void* ws, size_t ws_bytes, iseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ISEG_HIP_H */
