"""ctypes binding of libiseg_hip.so (the C ABI declared in include/iseg_hip.h).

This is the only place Python touches the native library.  There is deliberately NO fallback: if the shared
object is missing or a call fails, an exception is raised (HipLibraryMissing / HipCallError) -- the product
path never silently routes through torch ops or the oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libiseg_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_GELU_GRAD, ACT_RELU_GRAD, ACT_MUL_AUX, ACT_SIGMOID, ACT_SWISH = 0, 1, 2, 3, 4, 5, 6, 7


class HipLibraryMissing(RuntimeError):
    pass


class HipCallError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64), ("a_kcontig", C.c_int),
        ("B", C.c_void_p), ("ldb", C.c_int64), ("b_kcontig", C.c_int),
        ("D", C.c_void_p), ("ldd", C.c_int64),
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("in_dtype", C.c_int), ("out_dtype", C.c_int),
        ("bias", C.c_void_p), ("colscale", C.c_void_p), ("rowscale", C.c_void_p), ("rows_per_group", C.c_int64),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("pre_out", C.c_void_p), ("ldp", C.c_int64),
        ("act", C.c_int), ("alpha", C.c_float), ("accumulate", C.c_int), ("split_k", C.c_int), ("a_act", C.c_int),
        ("colsum_out", C.c_void_p), ("colsum_accumulate", C.c_int), ("defer_reduce", C.c_int),
        ("batch", C.c_int), ("batch_inner", C.c_int),
        ("sa_outer", C.c_int64), ("sa_inner", C.c_int64), ("sb_outer", C.c_int64), ("sb_inner", C.c_int64),
        ("sd_outer", C.c_int64), ("sd_inner", C.c_int64),
        ("pre_deriv", C.c_int),
        ("b_group_rows", C.c_int64),
        ("b_group_stride", C.c_int64),
        ("bias_rowscaled", C.c_int),
    ]


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "Cin", "Cout", "KH", "KW", "sh", "sw", "dh", "dw", "pt", "pl", "Ho", "Wo", "groups")]


_p, _i, _l, _f, _z, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); must list every symbol include/iseg_hip.h declares (tests/test_abi.py checks it)
SIGNATURES = {
    "iseg_version": (_i, []),
    "iseg_last_error": (_z, [C.c_char_p, _z]),
    "iseg_gemm_splits": (_i, [C.POINTER(GemmArgs)]),
    "iseg_gemm_tn_pair_splits": (_i, [C.POINTER(GemmArgs), C.POINTER(GemmArgs)]),
    "iseg_gemm_tn_pair": (_i, [C.POINTER(GemmArgs), _p, _z, C.POINTER(GemmArgs), _p, _z, _p]),
    "iseg_gemm_slabs": (_i, [C.POINTER(GemmArgs)]),
    "iseg_gemm_variant": (_i, [C.POINTER(GemmArgs)]),
    "iseg_gemm_workspace_bytes": (_z, [C.POINTER(GemmArgs)]),
    "iseg_gemm": (_i, [C.POINTER(GemmArgs), _p, _z, _p]),
    "iseg_gemm_reduce": (_i, [C.POINTER(GemmArgs), _p, _z, _p]),
    "iseg_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _f, _i, _p]),
    "iseg_layernorm_bwd_workspace_bytes": (_z, [_l, _i]),
    "iseg_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _z, _p]),
    "iseg_layernorm_post_fwd": (_i, [_p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _l, _i, _f, _i, _p]),
    "iseg_layernorm_post_bwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _l, _i, _i, _p, _z, _p]),
    "iseg_layernorm_gather_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _f, _i, _p]),
    "iseg_layernorm_gather_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _z, _p]),
    "iseg_dwconv2d_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "iseg_dcnv2_sample_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "iseg_dcnv2_sample_bwd_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "iseg_dcnv2_sample_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "iseg_qkv_rope": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _p]),
    "iseg_glu_fwd": (_i, [_p, _l, _p, _l, _p, _l, _l, _i, _i, _i, _p]),
    "iseg_glu_bwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _l, _i, _i, _i, _p]),
    "iseg_dwconv2d7_mfma": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "iseg_dwconv2d_bwd_weight_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "iseg_dwconv2d_bwd_weight": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "iseg_dwconv2d_strided_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "iseg_dwconv2d_strided_bwd_data": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "iseg_dwconv2d_strided_bwd_weight_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "iseg_dwconv2d_strided_bwd_weight": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "iseg_dcn_center_blend_fwd": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _p]),
    "iseg_dcn_center_blend_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p]),
    "iseg_bn_workspace_bytes": (_z, [_l, _i]),
    "iseg_bn_stats": (_i, [_p, _l, _p, _l, _i, _i, _p, _z, _p]),
    "iseg_bn_finalize": (_i, [_p, _i, _f, _f, _p, _p, _p, _p, _p]),
    "iseg_bn_apply_fwd": (_i, [_p, _l, _p, _p, _p, _p, _p, _l, _l, _i, _i, _i, _p]),
    "iseg_bn_apply_fwd_packed": (_i, [_p, _l, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _l, _l, _i, _i, _i, _p]),
    "iseg_bn_bwd_reduce": (_i, [_p, _l, _p, _l, _p, _l, _p, _p, _p, _l, _i, _i, _i, _p, _z, _p]),
    "iseg_bn_bwd_apply": (_i, [_p, _l, _p, _l, _p, _l, _p, _p, _p, _p, _f, _p, _l, _l, _i, _i, _i, _p]),
    "iseg_bn_bwd_apply_acc": (_i, [_p, _l, _p, _l, _p, _l, _p, _p, _p, _p, _f, _p, _l, _p, _p, _l, _i, _i, _i, _p]),
    "iseg_bn_bwd_reduce_remask": (_i, [_p, _l, _p, _l, _p, _p, _p, _p, _p, _l, _i, _i, _p, _z, _p]),
    "iseg_bn_bwd_apply_remask": (_i, [_p, _l, _p, _l, _p, _p, _p, _p, _p, _f, _p, _l, _p, _p, _l, _i, _i, _p]),
    "iseg_bn_relu_upsample_add": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "iseg_rsqrt_eps": (_i, [_p, _f, _p, _i, _p]),
    "iseg_cast": (_i, [_p, _i, _p, _i, _l, _p]),
    "iseg_deferred_begin": (_i, [_p, _z, _p, _z, _p]),
    "iseg_deferred_flush": (_i, [_p]),
    "iseg_deferred_end": (_i, [_p]),
    "iseg_transpose_batched": (_i, [_p, _p, _p, _i, _i, _p]),
    "iseg_scale_cols_cast": (_i, [_p, _p, _p, _l, _i, _i, _p]),
    "iseg_im2col": (_i, [_p, _i, _p, _i] + [_i] * 14 + [_l, _p]),
    "iseg_col2im": (_i, [_p, _p] + [_i] * 14 + [_l, _i, _p]),
    "iseg_colsum_workspace_bytes": (_z, [_i, _l, _i]),
    "iseg_colsum": (_i, [_p, _l, _l, _i, _l, _i, _p, _f, _i, _i, _p, _z, _p]),
    "iseg_broadcast_rows": (_i, [_p, _i, _p, _l, _l, _i, _l, _i, _f, _i, _i, _p]),
    "iseg_comm_unique_id": (_i, [_p]),
    "iseg_comm_init": (_i, [C.POINTER(C.c_void_p), _i, _i, _p]),
    "iseg_allreduce_sum": (_i, [_p, _p, _z, _i, _p]),
    "iseg_comm_destroy": (_i, [_p]),
    "iseg_axpby": (_i, [_p, _p, _p, _f, _f, _l, _i, _p]),
    "iseg_accumulate_pair": (_i, [_p, _i, _p, _p, _p]),
    "iseg_scale_dev": (_i, [_p, _p, _p, _l, _i, _p]),
    "iseg_rowscale": (_i, [_p, _p, _p, _l, _i, _l, _i, _p]),
    "iseg_dropout": (_i, [_p, _p, _l, _f, _u64, _p, _i, _p]),
    "iseg_drop_path_mask": (_i, [_p, _i, _f, _u64, _p, _p]),
    "iseg_drop_path_masks": (_i, [_p, _p, _i, _i, _u64, _p, _p]),
    "iseg_fill_f32": (_i, [_p, _f, _l, _p]),
    "iseg_act_fwd": (_i, [_p, _p, _l, _i, _i, _p]),
    "iseg_act_bwd": (_i, [_p, _p, _p, _l, _i, _i, _p]),
    "iseg_copy2d": (_i, [_p, _l, _p, _l, _l, _i, _i, _p]),
    "iseg_add2d_f32": (_i, [_p, _l, _p, _l, _l, _l, _p]),
    "iseg_scale_rows_f32": (_i, [_p, _p, _p, _l, _i, _p]),
    "iseg_layerscale_grads_workspace_bytes": (_z, [_i, _i]),
    "iseg_layerscale_grads": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "iseg_layerscale_grads_slabs": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "iseg_layerscale_grads_slabs_reduce": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p, _i, _l, _p, _p, _l, _i, _p]),
    "iseg_layerscale_grads_slabs_srow": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p, _i, _l, _p, _p, _l, _i, _p, _l, _l, _p, _l, _p]),
    "iseg_replace_nan_or_inf": (_i, [_p, _p, _l, _f, _i, _p, _z, _p]),
    "iseg_replace_nan_or_inf_bwd": (_i, [_p, _p, _p, _l, _i, _p]),
    "iseg_groupnorm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "iseg_groupnorm_bwd_workspace_bytes": (_z, [_i, _i]),
    "iseg_groupnorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "iseg_rmsnorm_fwd": (_i, [_p, _p, _p, _p, _l, _i, _f, _i, _p]),
    "iseg_rmsnorm_bwd_workspace_bytes": (_z, [_l, _i]),
    "iseg_rmsnorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _z, _p]),
    "iseg_grn_workspace_bytes": (_z, [_l, _l, _i]),
    "iseg_grn_fwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _l, _i, _f, _i, _p, _z, _p]),
    "iseg_grn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _l, _i, _f, _i, _p, _z, _p]),
    "iseg_grn_fold_weights": (_i, [_p, _p, _p, _p, _l, _i, _i, _p]),
    "iseg_grn_fold_bias": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "iseg_grn_fold_wgrad": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p]),
    "iseg_grn_bwd_folded": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _l, _i, _f, _i, _p, _z, _p]),
    "iseg_softmax_rows_fwd": (_i, [_p, _p, _l, _i, _i, _i, _p, _i, _p, _i, _f, _f, _i, _p]),
    "iseg_softmax_rows_bwd": (_i, [_p, _p, _p, _l, _i, _i, _f, _f, _i, _p]),
    "iseg_clip_fwd": (_i, [_p, _p, _l, _f, _f, _i, _p]),
    "iseg_clip_bwd": (_i, [_p, _p, _p, _l, _f, _f, _i, _p]),
    "iseg_gather_rows": (_i, [_p, _p, _p, _l, _l, _i, _i, _p]),
    "iseg_gather_rows_fma": (_i, [_p, _p, _p, _l, _i, _p, _p, _l, _i, _i, _p]),
    "iseg_relpos_bias_scatter_grad_window": (_i, [_p, _i, _p, _i, _i, _i, _p]),
    "iseg_colsum_wide_workspace_bytes": (_z, [_l, _l]),
    "iseg_colsum_wide": (_i, [_p, _l, _l, _l, _p, _i, _i, _p, _z, _p]),
    "iseg_relpos_bias_gather": (_i, [_p, _p, _p, _i, _i, _p]),
    "iseg_relpos_bias_scatter_grad": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "iseg_dcnv3_fwd": (_i, [_p, _p, _p, _p] + [_i] * 10 + [_f, _i, _p]),
    "iseg_dcnv3_bwd_workspace_bytes": (_z, [_i] * 10 + [_f]),
    "iseg_dcnv3_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p] + [_i] * 10 + [_f, _i, _p, _z, _p]),
    "iseg_dcnv3_fwd_ld": (_i, [_p, _p, _p, _l, _l, _p] + [_i] * 10 + [_f, _i, _p]),
    "iseg_dcnv3_bwd_side_bytes": (_z, [_i] * 10 + [_f]),
    "iseg_dcnv3_bwd_ld": (_i, [_p, _p, _p, _l, _l, _p, _p, _i, _p, _p] + [_i] * 10 + [_f, _i, _p, _z, _p, _z, _p]),
    "iseg_dcn_mask_softmax_fwd": (_i, [_p, _l, _i, _i, _l, _i, _i, _p]),
    "iseg_dcn_mask_softmax_bwd": (_i, [_p, _p, _l, _i, _i, _l, _i, _i, _p]),
    "iseg_split_cols_accumulate": (_i, [_p, _l, _l, _p, _i, _p, _i, _p]),
    "iseg_scale_cols": (_i, [_p, _p, _p, _l, _i, _i, _p]),
    "iseg_mul_colsum_workspace_bytes": (_z, [_l, _i]),
    "iseg_mul_colsum": (_i, [_p, _p, _l, _i, _p, _i, _i, _p, _z, _p]),
    "iseg_window_attention_supported": (_i, [_i, _i, _i]),
    "iseg_attention_fwd_supported": (_i, [_i, _i]),
    "iseg_attention_fwd": (_i, [_p, _p, _l, _i, _i, _i, _f, _i, _p]),
    "iseg_attention_lse_elems": (_z, [_l, _i, _i]),
    "iseg_attention_fwd_train": (_i, [_p, _p, _p, _l, _i, _i, _i, _f, _i, _p]),
    "iseg_attention_bwd_workspace_bytes": (_z, [_l, _i, _i]),
    "iseg_attention_bwd": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _i, _f, _i, _p, _z, _p]),
    "iseg_window_attention_table": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "iseg_window_attention_fwd": (_i, [_p, _p, _p, _l, _i, _i, _i, _f, _i, _p]),
    "iseg_window_attention_bwd_workspace_bytes": (_z, [_l, _i, _i]),
    "iseg_window_attention_bwd": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _i, _f, _i, _p, _z, _p]),
    "iseg_add_relu": (_i, [_p, _p, _p, _l, _i, _p]),
    "iseg_pool2d_fwd": (_i, [_p, _p] + [_i] * 14 + [_p]),
    "iseg_pool2d_bwd_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "iseg_pool2d_bwd": (_i, [_p, _p, _p] + [_i] * 14 + [_p, _z, _p]),
    "iseg_resize_bilinear_fwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "iseg_resize_bilinear_bwd_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "iseg_resize_bilinear_bwd": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "iseg_resize_bilinear_ac_fwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "iseg_resize_bilinear_ac_bwd": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "iseg_resize_nearest_i32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "iseg_softmax_ce_workspace_bytes": (_z, [_l, _i]),
    "iseg_softmax_ce_ignore": (_i, [_p, _p, _p, _l, _i, _i, _p, _p, _f, _p, _f, _p, _p, _z, _p]),
    "iseg_softmax_ce_confusion": (_i, [_p, _p, _p, _l, _i, _i, _p, _p, _f, _p, _f, _p, _p, _p, _z, _p]),
    "iseg_softmax_focal_ce_ignore": (_i, [_p, _p, _p, _l, _i, _i, _f, _f, _p, _p, _f, _p, _f, _p, _p, _z, _p]),
    "iseg_argmax_confusion": (_i, [_p, _p, _l, _i, _i, _p, _p, _p]),
    "iseg_grad_sqnorm": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _l, _i, _p]),
    "iseg_adamw_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _f, _p, _f, _p, _f, _l, _p]),
    "iseg_augment_params_ints": (_i, []),
    "iseg_augment_params_floats": (_i, []),
    "iseg_augment_means_workspace_bytes": (_z, [_i]),
    "iseg_augment_channel_means": (_i, [_p, _i, _p, _p, _i, _i, _i, _p, _z, _p]),
    "iseg_augment_crop_batch": (_i, [_p, _i, _p, _p, _p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _p, _p, _i, _i, _i,
                                     _i, _i, _u64, _p]),
    "iseg_normalize_image": (_i, [_p, _p, _l, C.POINTER(C.c_float), C.POINTER(C.c_float), _p]),
    "iseg_upsample_ce_supported": (_i, [_i, _i, _i, _i, _i]),
    "iseg_upsample_ce_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "iseg_upsample_ce": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _f, _p, _f, _p, _p, _z, _p]),
    "iseg_conv2d_igemm_supported": (_i, [C.POINTER(ConvGeom), _i]),
    "iseg_conv2d_igemm_workspace_bytes": (_z, [C.POINTER(ConvGeom), _i]),
    "iseg_conv2d_igemm_fwd": (_i, [_p, _p, _p, _p, C.POINTER(ConvGeom), _i, _p, _z, _p]),
    "iseg_conv2d_igemm_fwd_kt": (_i, [_p, _p, _p, _p, C.POINTER(ConvGeom), _i, _p, _z, _p]),
    "iseg_conv2d_igemm_fwd_kt_supported": (_i, [C.POINTER(ConvGeom), _i]),
    "iseg_conv2d_igemm_bwd_data": (_i, [_p, _p, _p, C.POINTER(ConvGeom), _i, _p, _z, _p]),
    "iseg_conv2d_igemm_bwd_weight": (_i, [_p, _p, _p, _i, C.POINTER(ConvGeom), _i, _p, _z, _p]),
    "iseg_convnext_mlp_supported": (_i, [_i, _i]),
    "iseg_convnext_mlp_tiled_bytes": (_z, [_i, _i]),
    "iseg_convnext_mlp_prep": (_i, [_p, _p, _p, _p, _p, _i, _p]),
    "iseg_convnext_mlp_fwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _p, _p, _l, _i, _i, _p]),
    "iseg_convnext_mlp_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _p]),
    "iseg_convnext_mlp_bwd_data": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _l, _i, _i, _p]),
    "iseg_convnext_mlp_bwd_data_ln_workspace_bytes": (_z, [_l, _i]),
    "iseg_convnext_mlp_bwd_data_ln": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _l, _i, _i, _p, _z, _p]),
    "iseg_convnext_weight_prep_batched": (_i, [_p, _i, _l, _p]),
    "iseg_convnext_mlp_fwd_ln": (_i, [_p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _l, _i, _i, _p]),
    "iseg_convnext_mlp_wgrad_workspace_bytes": (_z, [_l, _i]),
    "iseg_convnext_mlp_wgrad": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _p, _z, _p]),
    "iseg_sgd_momentum_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _f, _i, _p, _f, _p, _f, _l, _p]),
}

_lib = None


def lib():
    """Load (once) and return the CDLL; raises HipLibraryMissing when the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m iseg_amd.build` (hipcc --offload-arch=gfx950). "
                "iseg_amd has no CPU or torch fallback.")
        # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1; the process must hold ONE HIP runtime, and the device
        # pointers this library receives come from torch's.  Loading torch first makes the dynamic linker resolve our DT_NEEDED
        # entries (same SONAMEs) to the copies torch already mapped; the other order leaves two runtimes in the process and every
        # launch from here fails with "no ROCm-capable device is detected" (seen when build() and smoke() shared one process).
        import torch  # noqa: F401

        dll = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(dll, name)
            fn.restype = res
            fn.argtypes = args
        _lib = dll
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().iseg_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(status, what):
    if status != 0:
        raise HipCallError(f"{what} failed with status {status}: {last_error()}")


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
