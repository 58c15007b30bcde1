"""ctypes binding of libmclstexp_hip.so (the C ABI in include/mclstexp_hip.h).

There is deliberately no fallback: if the library is missing or a call fails, a RuntimeError is
raised.  ``lib()`` loads lazily so that host-side code (argument parsing, module construction,
state-dict handling) can be imported on a machine without the library built.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch bundles its own libamdhip64.so.7; it MUST be in the process before our library is dlopen-ed so that
# both bind to the same HIP runtime instance (same SONAME as /opt/rocm's: whichever loads first wins).
# Loading ours first would give it a second, device-less runtime ("no ROCm-capable device", hipError 100).
import torch  # noqa: F401  (side effect: loads torch/lib/libamdhip64.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCL_LIB_PATH") or os.path.join(_HERE, "libmclstexp_hip.so")   # override: A/B of kernel builds
ABI_VERSION = 9

_lib: Optional[C.CDLL] = None

c_f = C.c_float
c_d = C.c_double
c_i = C.c_int32
c_l = C.c_int64
c_p = C.c_void_p


class GemmArgs(C.Structure):
    """struct mcl_gemm_args (include/mclstexp_hip.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("M", c_i), ("N", c_i), ("K", c_i), ("batch", c_i),
        ("A", c_p), ("sAm", c_l), ("sAk", c_l), ("sAb", c_l),
        ("B", c_p), ("sBk", c_l), ("sBn", c_l), ("sBb", c_l),
        ("C", c_p), ("ldc", c_l), ("sCb", c_l),
        ("alpha", c_f), ("flags", c_i),
        ("bias", c_p),
        ("resid", c_p), ("ldr", c_l), ("sRb", c_l),
        ("pre_out", c_p), ("ldp", c_l),
        ("aux", c_p), ("ldaux", c_l),
        ("compute", c_i), ("ksplit", c_i),
        ("workspace", c_p),
        ("flt_thr", c_p), ("flt_cnt", c_p), ("flt_val", c_p), ("flt_idx", c_p), ("flt_cap", c_i),
        ("counters", c_p),
    ]




def gemm_args(**kw) -> GemmArgs:
    """A zeroed ``mcl_gemm_args`` with ``struct_size`` filled in (the library ignores fields beyond it)."""
    a = GemmArgs(**kw)
    a.struct_size = C.sizeof(GemmArgs)
    return a


EPI_GELU, EPI_GELU_BWD, EPI_ACCUM = 1, 2, 4
COMPUTE_F32, COMPUTE_BF16 = 0, 1

# name -> argtypes (restype is int unless listed in _RESTYPES)
PROTOTYPES = {
    "mcl_abi_version": [],
    "mcl_error_string": [c_i],
    "mcl_gemm": [C.POINTER(GemmArgs), c_p],
    "mcl_gemm_args_size": [],
    "mcl_gemm_args_min_size": [],
    "mcl_gemm_auto_ksplit": [c_i, c_i, c_i, c_i],
    "mcl_gemm_workspace_floats": [c_i, c_i, c_i, c_i],
    "mcl_gemm_group": [C.POINTER(GemmArgs), c_i, c_p],
    "mcl_proj_head_ksplit": [c_i, c_i],
    "mcl_proj_head_ws_floats": [c_i, c_i],
    "mcl_proj_head_fwd": [c_p, c_l, c_i, c_i, c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                          c_i, c_p],
    "mcl_proj_head_bwd_rows": [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p],
    "mcl_pos_embed_add_fwd": [c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_p],
    "mcl_embed_rowgrad": [c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_i, c_p],
    "mcl_embed_scatter_rows": [c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p],
    "mcl_layernorm_fwd": [c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_i, c_f, c_p],
    "mcl_layernorm_bwd": [c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_p],
    "mcl_attention_fwd": [c_p, c_l, c_i, c_i, c_i, c_f, c_p, c_l, c_p, c_p],
    "mcl_attention_bwd": [c_p, c_l, c_i, c_i, c_i, c_f, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_p],
    "mcl_attention_batched_fwd": [c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_p, c_l, c_p, c_p],
    "mcl_attention_batched_bwd": [c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_p],
    "mcl_softmax_rows_fwd": [c_p, c_l, c_i, c_i, c_f, c_p],
    "mcl_softmax_rows_bwd": [c_p, c_p, c_l, c_i, c_i, c_f, c_p],
    "mcl_colsum": [c_p, c_l, c_p, c_i, c_i, c_i, c_p],
    "mcl_colred_group": [c_i, C.POINTER(c_p), C.POINTER(c_l), C.POINTER(c_p), C.POINTER(c_l), C.POINTER(c_p), C.POINTER(c_p),
                         C.POINTER(c_p), C.POINTER(c_p), C.POINTER(c_i), C.POINTER(c_i), c_i, c_p],
    "mcl_soft_clip_mid": [c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p],
    "mcl_symmetrize": [c_p, c_i, c_p, c_p],
    "mcl_colsum_ws": [c_p, c_l, c_p, c_i, c_i, c_i, c_p, c_p],
    "mcl_rowred_workspace_floats": [c_i, c_i],
    "mcl_layernorm_bwd_ws": [c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "mcl_infonce_lse": [c_p, c_l, c_i, c_i, c_p, c_p, c_p],
    "mcl_infonce_loss": [c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "mcl_infonce_dlogits": [c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_l, c_p],
    "mcl_infonce_fused_workspace_bytes": [c_i, c_i, c_i],
    "mcl_infonce_fused_lse": [c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_l, c_p],
    "mcl_infonce_fused_grad": [c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_f, c_p, c_p, c_l, c_p],
    "mcl_cast_f32_to_bf16": [c_p, c_l, c_p, c_l, c_l, c_i, c_p],
    "mcl_infonce_loss_mean": [c_p, c_l, c_p, c_p, c_i, c_f, c_p, c_p],
    "mcl_adam_step_dev_shadow": [c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_p],
    "mcl_bn_gap_fwd": [c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "mcl_bn_gap_bwd": [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_p],
    "mcl_bn_running_update": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "mcl_bn_eval_rstd": [c_i, c_p, c_p, c_p, c_p, c_p],
    "mcl_image_to_bf16_nhwc": [c_p, c_l, c_l, c_l, c_l, c_i, c_i, c_i, c_i, c_p, c_p],
    "mcl_fill_zero": [c_p, c_l, c_p],
    "mcl_stamp": [c_p, c_i, c_p],
    "mcl_dropout_fwd": [c_p, c_p, c_p, c_l, c_f, C.c_uint64, c_p],
    "mcl_dropout_bwd": [c_p, c_p, c_p, c_l, c_f, c_p],
    "mcl_gelu_f32": [c_p, c_p, c_p, c_l, c_p],
    "mcl_add_f32": [c_p, c_p, c_p, c_l, c_p],
    "mcl_im2col_nhwc": [c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "mcl_col2im_nhwc": [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_i, c_p],
    "mcl_maxpool3s2_nhwc_fwd_any": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "mcl_maxpool3s2_nhwc_bwd_any": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "mcl_avgpool2_nhwc_any": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "mcl_gap_nhwc_fwd": [c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_p],
    "mcl_gap_nhwc_bwd": [c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "mcl_add_relu": [c_p, c_p, c_p, c_l, c_i, c_i, c_p],
    "mcl_scale2_f32": [c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p],
    "mcl_maxpool3s2_nhwc_bf16_bwd_ld": [c_p, c_p, c_l, c_p, c_i, c_i, c_i, c_i, c_p],
    "mcl_gemm_bf16_workspace_floats": [c_i, c_l, c_i],
    "mcl_gemm_bf16": [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_f, c_i, c_p,
                      c_p, c_l, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_i, c_p],
    "mcl_ln_bf16_fwd": [c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_l, c_i, c_f, c_p],
    "mcl_colred_workspace_floats": [c_l, c_i],
    "mcl_ln_bf16_bwd": [c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_i, c_l, c_i, c_p],
    "mcl_colsum_bf16": [c_p, c_l, c_l, c_i, c_p, c_p, c_i, c_p],
    "mcl_softmax_bf16_fwd": [c_p, c_l, c_l, c_i, c_p],
    "mcl_softmax_bf16_bwd": [c_p, c_p, c_l, c_l, c_i, c_f, c_p],
    "mcl_vit_patchify": [c_p, c_l, c_l, c_l, c_l, c_i, c_i, c_i, c_i, c_p, c_p],
    "mcl_vit_patchify_tokens": [c_p, c_l, c_l, c_l, c_l, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p],
    "mcl_vit_cls_row": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "mcl_vit_assemble_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "mcl_vit_tokens_extract": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "mcl_vit_token_mean_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "mcl_vit_token_mean_bwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "mcl_vit_pos_grad": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "mcl_vit_zero_cls_rows": [c_p, c_i, c_i, c_i, c_i, c_p],
    "mcl_strided4_f32": [c_p, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_l, c_i, c_i, c_p],
    "mcl_copy_rows": [c_p, c_l, c_p, c_l, c_l, c_l, c_p],
    "mcl_weight_rot180": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "mcl_vit_attn_fwd": [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "mcl_vit_attn_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "mcl_quant_e4m3_rows": [c_p, c_l, c_i, c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_p],
    "mcl_dequant_e4m3_rows": [c_p, c_l, c_p, c_l, c_i, c_i, c_p, c_l, c_p],
    "mcl_infonce_fp8_workspace_bytes": [c_i, c_i],
    "mcl_infonce_fp8_lse": [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_f, c_p, c_p, c_l, c_p],
    "mcl_infonce_rowdot_bf16": [c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_p, c_p],
    "mcl_bn_workspace_floats": [c_l, c_i, c_i],
    "mcl_bn_stats": [c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_p, c_f, c_p, c_p, c_p, c_p],
    "mcl_bn_act_fwd": [c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_p],
    "mcl_bn_act_bwd": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_p],
    "mcl_dense_conv1x1_workspace_floats": [c_l],
    "mcl_dense_conv1x1_fwd": [c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_f, c_p, c_p, c_p, c_p],
    "mcl_dense_conv3x3_workspace_floats": [c_l],
    "mcl_dense_bn1_bwd_workspace_floats": [c_l, c_i],
    "mcl_dense_conv3x3_bwd_workspace_floats": [c_l],
    "mcl_avgpool2_nhwc_bf16": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "mcl_maxpool3s2_nhwc_bf16_fwd": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "mcl_maxpool3s2_nhwc_bf16_bwd": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "mcl_dense_conv3x3_bwd": [c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p],
    "mcl_dense_conv3x3_bwd_fix": [c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p,
                                  c_l, c_p, c_p, c_p, c_p, c_p],
    "mcl_dense_bn1_bwd": [c_p, c_p, c_i, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_p],
    "mcl_dense_conv3x3_wrw_workspace_floats": [c_l],
    "mcl_dense_conv3x3_wrw_det": [c_p, c_l, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    "mcl_wrw_workspace_floats": [c_l, c_i, c_i],
    "mcl_dense_bn1_wrw": [c_p, c_p, c_i, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p],
    "mcl_dense_bn1_dx_window": [c_p, c_p, c_i, c_i, c_i, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p],
    "mcl_dense_bn1_dx_pair": [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_l, c_p, c_p, c_p, c_l, c_p],
    "mcl_dense_bn1_dx": [c_p, c_p, c_i, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p],
    "mcl_dense_bn1_dx_sums": [c_p, c_p, c_i, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_l, c_p],
    "mcl_dense_bn1_fix": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_p, c_p],
    "mcl_conv1x1_wrw_det": [c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_p],
    "mcl_dense_conv3x3_fwd": [c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_f, c_p, c_p, c_p, c_p],
    "mcl_accum_into_f32": [c_p, c_p, c_l, c_i, c_p],
    "mcl_adam_step": [c_p, c_p, c_p, c_p, c_l, c_d, c_d, c_d, c_d, c_d, c_d, c_d, c_p],
    "mcl_adam_table_step": [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_l, c_d, c_d, c_d, c_d, c_d, c_d, c_d, c_p],
    "mcl_adam_consts_update": [c_p, c_p, c_p, c_p],
    "mcl_adam_step_dev": [c_p, c_p, c_p, c_p, c_l, c_p, c_p],
    "mcl_adam_table_step_dev": [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_l, c_p, c_p],
    "mcl_row_slot_update": [c_p, c_p, c_i, c_i, c_p],
    "mcl_dense_block_fwd_workspace_bytes": [c_i, c_i],
    "mcl_dense_block_pack_w1": [c_p, c_p, c_i, c_i, c_p],
    "mcl_dense_block_pack_bwd": [c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    "mcl_dense_block_bwd_workspace_bytes": [c_i, c_i],
    "mcl_dense_block_bwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, C.c_uint32, c_p, c_p],
    "mcl_dense_block_fwd": [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, C.c_uint32, c_p, c_p],
    "mcl_adam_consts_update_hist": [c_p, c_p, c_p, c_p, c_i, c_p],
    "mcl_adam_table_lazy": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_p, c_p, c_i,
                            c_p],
    "mcl_bn_act_avgpool_fwd": [c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_p],
    "mcl_bn_act_avgpool_bwd": [c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l,
                               c_p],
    "mcl_conv0_workspace_floats": [c_i, c_i, c_i],
    "mcl_conv0_fwd": [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p],
    "mcl_conv0_wrw_workspace_floats": [c_i, c_i, c_i],
    "mcl_conv0_wrw": [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p],
    "mcl_bn_act_maxpool_fwd": [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "mcl_patch_gather": [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_f, c_p, c_p, c_p],
    "mcl_her2st_train_patches": [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_f, c_p, c_p, c_p],
    "mcl_log_library_size_normalize": [c_p, c_l, c_p, c_l, c_i, c_i, c_f, c_p],
    "mcl_l2_normalize_rows": [c_p, c_l, c_p, c_l, c_i, c_i, c_p],
    "mcl_topk_rows_max_k": [],
    "mcl_topk_rows": [c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p],
    "mcl_topk_rows_indexed": [c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "mcl_knn_weighted_average": [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
}
_RESTYPES = {"mcl_error_string": C.c_char_p, "mcl_gemm_args_size": C.c_uint32, "mcl_gemm_args_min_size": C.c_uint32, "mcl_dense_block_fwd_workspace_bytes": C.c_int64, "mcl_dense_block_bwd_workspace_bytes": C.c_int64, "mcl_bn_workspace_floats": C.c_int64,
             "mcl_infonce_fused_workspace_bytes": C.c_int64, "mcl_dense_conv1x1_workspace_floats": C.c_int64,
             "mcl_dense_conv3x3_workspace_floats": C.c_int64, "mcl_dense_bn1_bwd_workspace_floats": C.c_int64,
             "mcl_dense_conv3x3_bwd_workspace_floats": C.c_int64, "mcl_conv0_workspace_floats": C.c_int64,
             "mcl_wrw_workspace_floats": C.c_int64, "mcl_dense_conv3x3_wrw_workspace_floats": C.c_int64,
             "mcl_conv0_wrw_workspace_floats": C.c_int64, "mcl_infonce_fp8_workspace_bytes": C.c_int64,
             "mcl_gemm_bf16_workspace_floats": C.c_int64, "mcl_colred_workspace_floats": C.c_int64,
             "mcl_gemm_workspace_floats": C.c_int64, "mcl_rowred_workspace_floats": C.c_int64,
             "mcl_proj_head_ws_floats": C.c_int64}


def load(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the library and attach prototypes.  Raises RuntimeError if absent or ABI-mismatched."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built.  Run `python -m mclstexp_amd.build` "
            "(hipcc --offload-arch=gfx950).  mclstexp_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, argtypes in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{path} does not export {name}; rebuild the library") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    v = lib.mcl_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"{path} has ABI version {v}, host expects {ABI_VERSION}; rebuild")
    if lib.mcl_gemm_args_size() != C.sizeof(GemmArgs):
        raise RuntimeError(f"{path}: sizeof(mcl_gemm_args) = {lib.mcl_gemm_args_size()}, the ctypes declaration in "
                           f"mclstexp_amd/_lib.py has {C.sizeof(GemmArgs)}; the binding and include/mclstexp_hip.h drifted")
    return lib


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = lib().mcl_error_string(code)
        raise RuntimeError(f"mclstexp_hip {what} failed: [{code}] {msg.decode() if msg else '?'}")


class AbiTimer:
    """Times selected C-ABI entry points with HIP events recorded on the stream they launch on.

    ``with AbiTimer(["mcl_dense_bn1_wrw", ...]) as t: step()`` wraps each named entry point so that every call is
    bracketed by two timing events on torch's CURRENT stream -- the stream ``ops._stream()`` hands to the ABI, i.e.
    the launch stream.  ``t.summary()`` (after a device synchronise) returns, per name, the call count, the mean and
    total duration in ms, and the recorded argument tuples (bench.py derives each call's algorithmic bytes from
    them).  An entry point that enqueues more than one kernel (a reduce + its finalize) is timed as one unit."""

    def __init__(self, names):
        self.names = list(names)
        self.records = {n: [] for n in self.names}
        self._orig = {}

    def __enter__(self):
        L = lib()
        for n in self.names:
            orig = getattr(L, n)
            self._orig[n] = orig

            def wrapped(*args, _orig=orig, _n=n):
                st = torch.cuda.current_stream()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                rc = _orig(*args)
                e1.record(st)
                self.records[_n].append((e0, e1, args))
                return rc
            setattr(L, n, wrapped)
        return self

    def __exit__(self, *exc):
        L = lib()
        for n, orig in self._orig.items():
            setattr(L, n, orig)
        self._orig = {}
        return False

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for n, recs in self.records.items():
            if not recs:
                continue
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[n] = {"calls": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms), "ms": ms,
                      "args": [r[2] for r in recs]}
        return out
