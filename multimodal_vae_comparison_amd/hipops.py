"""ctypes binding of the C-ABI hot-path library (include/mmvae_hip.h -> libmmvae_hip.so).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every function below passes
raw device pointers + sizes + the stream to the extern "C" entry points.  There is NO fallback: if the
library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmvae_hip.so")
_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_long
c_f = ctypes.c_float
c_u = ctypes.c_uint
c_sz = ctypes.c_size_t

MAX_EXPERTS = 8

ACT_NONE, ACT_SILU, ACT_RELU, ACT_GELU = 0, 1, 2, 3
EP_NONE, EP_RELU, EP_MUL_RELU_MASK, EP_MUL_SILU_GRAD, EP_GELU, EP_MUL_GELU_GRAD, EP_SIGMOID_CLAMP, EP_SIGMOID, EP_ADD_AUX = range(9)

_ERR = {1: "invalid argument", 2: "unsupported shape", 3: "kernel launch failed"}


class PoeFwdArgs(ctypes.Structure):
    _fields_ = [("mu", c_p * MAX_EXPERTS), ("lv", c_p * MAX_EXPERTS), ("eps", c_p * MAX_EXPERTS),
                ("z", c_p * MAX_EXPERTS)]


class PoeBwdArgs(ctypes.Structure):
    _fields_ = [("mu", c_p * MAX_EXPERTS), ("lv", c_p * MAX_EXPERTS), ("eps", c_p * MAX_EXPERTS),
                ("dz", c_p * MAX_EXPERTS), ("dmu", c_p * MAX_EXPERTS), ("dlv", c_p * MAX_EXPERTS)]


FAN_MAX_BLOCKS, FAN_MAX_SRC, FAN_MAX_USES = 16, 8, 4


class FanBlocks(ctypes.Structure):
    _fields_ = [("src", c_p * FAN_MAX_BLOCKS), ("dst", c_p * FAN_MAX_BLOCKS), ("width", c_i * FAN_MAX_BLOCKS),
                ("ld_src", c_i * FAN_MAX_BLOCKS), ("ld_dst", c_i * FAN_MAX_BLOCKS), ("n", c_i), ("B", c_i)]


class FanSum(ctypes.Structure):
    _fields_ = [("out", c_p * FAN_MAX_SRC), ("g", (c_p * FAN_MAX_USES) * FAN_MAX_SRC), ("ld", (c_i * FAN_MAX_USES) * FAN_MAX_SRC),
                ("n_g", c_i * FAN_MAX_SRC), ("width", c_i * FAN_MAX_SRC), ("n", c_i), ("B", c_i)]


class Dropout(ctypes.Structure):
    _fields_ = [("state", c_p), ("slot", c_u), ("site", c_u), ("p", c_f)]


MOE_MAX_MODS = 4


class MoeKArgs(ctypes.Structure):
    _fields_ = [("packed", c_p * MOE_MAX_MODS), ("eps", c_p * MOE_MAX_MODS), ("z", c_p * MOE_MAX_MODS),
                ("laplace", c_i * MOE_MAX_MODS)]


class MoeKBwdArgs(ctypes.Structure):
    _fields_ = [("packed", c_p * MOE_MAX_MODS), ("eps", c_p * MOE_MAX_MODS), ("z", c_p * MOE_MAX_MODS),
                ("dz", c_p * MOE_MAX_MODS), ("dpacked", c_p * MOE_MAX_MODS), ("laplace", c_i * MOE_MAX_MODS)]


class DregRows(ctypes.Structure):      # also mmvae_dreg_rows_grad (same layout, non-const pointers)
    _fields_ = [("own", c_p * MOE_MAX_MODS), ("cross", c_p * MOE_MAX_MODS), ("lam", c_f * MOE_MAX_MODS)]


class RcStat(ctypes.Structure):      # mmvae_rc_stat_t: a BatchNorm whose backward statistics a kernel produces
    _fields_ = [("Y", c_p), ("mean", c_p), ("rstd", c_p), ("gamma", c_p), ("pqr", c_p), ("dgamma", c_p), ("dbeta", c_p),
                ("part", c_p), ("counter", c_p), ("acc", c_i), ("eval", c_i)]


class RcGeom(ctypes.Structure):      # mmvae_rc_geom_t
    _fields_ = [("H", c_i), ("W", c_i), ("Ho", c_i), ("Wo", c_i), ("KW", c_i), ("S", c_i), ("P", c_i)]


class RcFwd(ctypes.Structure):       # mmvae_rc_fwd_t
    _fields_ = [("x", c_p), ("w", c_p), ("xmean", c_p), ("xsc", c_p), ("xbeta", c_p), ("y", c_p), ("ws", c_p),
                ("tile_ticket", c_p), ("M", c_i), ("Cin", c_i), ("Cout", c_i), ("T", c_i), ("pre", c_i), ("g", RcGeom),
                ("gamma", c_p),
                ("beta", c_p), ("run_mean", c_p), ("run_var", c_p), ("mean", c_p), ("rstd", c_p), ("sc", c_p), ("part", c_p),
                ("counter", c_p), ("eps", c_f), ("momentum", c_f), ("eval", c_i)]


class RcDgrad(ctypes.Structure):     # mmvae_rc_dgrad_t
    _fields_ = [("G", c_p), ("Y", c_p), ("pqr", c_p), ("w", c_p), ("add", c_p), ("add_tbl", c_p), ("mask", c_i),
                ("mY", c_p), ("mmean", c_p), ("msc", c_p), ("mbeta", c_p), ("out", c_p), ("ws", c_p), ("tile_ticket", c_p),
                ("row_map", c_p), ("M", c_i), ("Min", c_i), ("Cin", c_i), ("Cout", c_i), ("T", c_i), ("nstat", c_i), ("g", RcGeom),
                ("st", RcStat * 2)]


class RcWgrad(ctypes.Structure):     # mmvae_rc_wgrad_t
    _fields_ = [("G", c_p), ("Y", c_p), ("pqr", c_p), ("x", c_p), ("xmean", c_p), ("xsc", c_p), ("xbeta", c_p), ("tbl", c_p),
                ("dw", c_p), ("ws", c_p), ("counter", c_p), ("M", c_i), ("Cin", c_i), ("Cout", c_i), ("T", c_i), ("pre", c_i),
                ("accumulate", c_i)]


RC_MAX_JOBS = 8


class RcJob(ctypes.Structure):       # mmvae_rc_job_t
    _fields_ = [("kind", c_i), ("f", RcFwd), ("d", RcDgrad), ("w", RcWgrad)]


c_dp = ctypes.POINTER(Dropout)
DROPOUT_SLOTS = 16

# name -> (restype, argtypes); must list every symbol include/mmvae_hip.h declares (tests check this)
SIGNATURES = {
    "mmvae_version": (c_i, []),
    "mmvae_arch": (ctypes.c_char_p, []),
    "mmvae_conv2d_k4s2_fwd": (c_i, [c_p] * 5 + [c_i] * 6 + [c_p]),
    "mmvae_conv2d_k4s2_dgrad": (c_i, [c_p] * 4 + [c_i] * 5 + [c_p]),
    "mmvae_conv2d_k4s2_wgrad": (c_i, [c_p] * 5 + [c_i] * 6 + [c_p]),
    "mmvae_convT2d_k4s2_fwd": (c_i, [c_p] * 5 + [c_i] * 6 + [c_p]),
    "mmvae_convT2d_k4s2_dgrad": (c_i, [c_p] * 4 + [c_i] * 5 + [c_p]),
    "mmvae_convT2d_k4s2_wgrad": (c_i, [c_p] * 5 + [c_i] * 6 + [c_p]),
    "mmvae_conv_wgrad_ws_floats": (c_sz, [c_i] * 4),
    "mmvae_conv2d_k4s2_bwd": (c_i, [c_p] * 7 + [c_i] * 6 + [c_p]),
    "mmvae_convT2d_k4s2_bwd": (c_i, [c_p] * 7 + [c_i] * 6 + [c_p]),
    "mmvae_gemm_f32": (c_i, [c_p] * 7 + [c_i] * 3 + [c_l] * 5 + [c_i] * 5 + [c_p]),
    "mmvae_gemm_ws_floats": (c_sz, [c_i] * 3),
    "mmvae_gemm_splits": (c_i, [c_i] * 4),
    "mmvae_bias_group_add": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "mmvae_bias_group_grad": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "mmvae_bias_group_ws_floats": (c_sz, [c_i, c_i]),
    "mmvae_bias_group_parts": (c_i, [c_i]),
    "mmvae_linear_fwd": (c_i, [c_p] * 5 + [c_i] * 3 + [c_l] + [c_i] * 2 + [c_p]),
    "mmvae_gemm_b16_set": (c_i, [c_i]),
    "mmvae_attn_t_bwd_set": (c_i, [c_i]),
    "mmvae_linear_bwd_data": (c_i, [c_p] * 4 + [c_i] * 5 + [c_p]),
    "mmvae_linear_bwd_weight": (c_i, [c_p] * 5 + [c_i] * 3 + [c_l] + [c_i] * 2 + [c_p]),
    "mmvae_linear_bwd_weight_ws_floats": (c_sz, [c_i] * 3),
    "mmvae_linear_bwd_weight_batch": (c_i, [c_p, c_i, c_p]),
    "mmvae_txt_layer_plan": (c_i, [c_i] * 3),
    "mmvae_conv_plan": (c_i, [c_i]),
    "mmvae_convT3_bce_seeded": (c_i, [c_p] * 8 + [c_i, c_i, c_f, c_p]),
    "mmvae_convT3_bce_strips": (c_i, [c_i]),
    "mmvae_txt_wgrad": (c_i, [c_p, c_i, c_p]),
    "mmvae_txt_wgrad_splits": (c_i, [c_i] * 3),
    "mmvae_txt_wgrad_ws_floats": (c_sz, [c_i] * 3),
    "mmvae_txt_wgrad_supported": (c_i, [c_i] * 3),
    "mmvae_input_pipe_create": (c_i, [c_p, c_p, c_sz]),
    "mmvae_input_pipe_destroy": (c_i, [c_p]),
    "mmvae_input_pipe_prefetch": (c_i, [c_p, c_p]),
    "mmvae_input_pipe_commit": (c_i, [c_p, c_p, c_i, c_p, c_p]),
    "mmvae_linear_bwd": (c_i, [c_p] * 8 + [c_i] * 3 + [c_l] + [c_i] * 3 + [c_p]),
    "mmvae_linear_bwd_ws_floats": (c_sz, [c_i] * 3),
    "mmvae_head_softmax_fwd": (c_i, [c_p, c_i, c_i, c_p]),
    "mmvae_head_softmax_bwd": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "mmvae_poe_reparam_kl_fwd": (c_i, [ctypes.POINTER(PoeFwdArgs), c_p, c_p, c_p, c_i, c_i, c_i, c_u, c_i, c_i, c_i,
                                       c_i, c_p, c_p]),
    "mmvae_poe_reparam_kl_bwd": (c_i, [ctypes.POINTER(PoeBwdArgs), c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_u, c_i, c_i,
                                       c_i, c_i, c_i, c_p]),
    "mmvae_poe_reparam_kl_bwd_acc": (c_i, [ctypes.POINTER(PoeBwdArgs), c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_u, c_i,
                                           c_i, c_i, c_i, c_i, c_u, c_p]),
    "mmvae_poe_ws_floats": (c_sz, [c_i, c_i]),
    "mmvae_bce_rowsum_fwd": (c_i, [c_p] * 3 + [c_i] * 3 + [c_p]),
    "mmvae_bce_sigmoid_clamp_bwd": (c_i, [c_p] * 4 + [c_i] * 3 + [c_p]),
    "mmvae_bce_rowsum_bwd": (c_i, [c_p] * 4 + [c_i] * 2 + [c_p]),
    "mmvae_sigmoid_clamp_bwd": (c_i, [c_p] * 3 + [c_l] + [c_p]),
    "mmvae_bce_elem_fwd": (c_i, [c_p] * 3 + [c_l] + [c_p]),
    "mmvae_bce_rowsum_seeded": (c_i, [c_p] * 3 + [c_f] + [c_p] + [c_i] * 2 + [c_p]),
    "mmvae_ce_over_time_seeded": (c_i, [c_p] * 3 + [c_f] + [c_p] + [c_i] * 3 + [c_p]),
    "mmvae_lprob_elem_fwd": (c_i, [c_p] * 3 + [c_l, c_l, c_f, c_i, c_p]),
    "mmvae_lprob_elem_bwd": (c_i, [c_p] * 4 + [c_l, c_l, c_f, c_i, c_p]),
    "mmvae_optimal_sigma_elem_fwd": (c_i, [c_p] * 5 + [c_l, c_p]),
    "mmvae_optimal_sigma_elem_bwd": (c_i, [c_p] * 6 + [c_l, c_p]),
    "mmvae_pointwise_rowsum_fwd": (c_i, [c_p] * 3 + [c_i] * 4 + [c_p]),
    "mmvae_pointwise_rowsum_bwd": (c_i, [c_p] * 4 + [c_i] * 4 + [c_p]),
    "mmvae_pointwise_elem": (c_i, [c_p] * 4 + [c_l, c_i, c_p]),
    "mmvae_ce_over_time_fwd": (c_i, [c_p] * 4 + [c_i] * 4 + [c_p]),
    "mmvae_ce_over_time_bwd": (c_i, [c_p] * 5 + [c_i] * 4 + [c_p]),
    "mmvae_lincomb_rows_fwd": (c_i, [c_p, ctypes.POINTER(c_f), c_p, c_i, c_i, c_i, c_p]),
    "mmvae_lincomb_rows_bwd": (c_i, [c_p, ctypes.POINTER(c_f), c_p, c_i, c_i, c_i, c_p]),
    "mmvae_expand_image_u8": (c_i, [c_p, c_p, c_l, c_p]),
    "mmvae_expand_text_tokens": (c_i, [c_p] * 4 + [c_i] * 3 + [c_p]),
    "mmvae_randn": (c_i, [c_p, c_l, c_p, c_p]),
    "mmvae_debug_timestamp": (c_i, [c_p, c_p]),
    "mmvae_debug_spin": (c_i, [c_p, ctypes.c_longlong, c_p]),
    "mmvae_conv2d_generic_fwd": (c_i, [c_p] * 5 + [c_i] * 10 + [c_p]),
    "mmvae_conv2d_generic_dgrad": (c_i, [c_p] * 4 + [c_i] * 9 + [c_p]),
    "mmvae_conv2d_generic_wgrad": (c_i, [c_p] * 4 + [c_i] * 10 + [c_p]),
    "mmvae_convT2d_generic_fwd": (c_i, [c_p] * 5 + [c_i] * 10 + [c_p]),
    "mmvae_convT2d_generic_dgrad": (c_i, [c_p] * 4 + [c_i] * 9 + [c_p]),
    "mmvae_convT2d_generic_wgrad": (c_i, [c_p] * 4 + [c_i] * 10 + [c_p]),
    "mmvae_sigmoid_fwd": (c_i, [c_p, c_p, c_l, c_p]),
    "mmvae_sigmoid_bwd": (c_i, [c_p, c_p, c_p, c_l, c_p]),
    "mmvae_lprob_rowsum_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_p]),
    "mmvae_lprob_rowsum_bwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_p]),
    "mmvae_optimal_sigma_ws_floats": (c_sz, [c_i, c_i]),
    "mmvae_optimal_sigma_fwd": (c_i, [c_p] * 5 + [c_i, c_i, c_p]),
    "mmvae_optimal_sigma_bwd": (c_i, [c_p] * 5 + [c_i, c_i, c_p]),
    "mmvae_add_pe_dropout_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_dp, c_p]),
    "mmvae_lincomb_rowptrs_fwd": (c_i, [c_p, ctypes.POINTER(c_f), c_p, c_p, c_i, c_i, c_i, c_p]),
    "mmvae_lincomb_rowptrs_bwd": (c_i, [c_p, ctypes.POINTER(c_f), c_p, c_i, c_i, c_i, c_p]),
    "mmvae_embed_pe_fwd": (c_i, [c_p] * 4 + [c_i] * 5 + [c_dp, c_p]),
    "mmvae_embed_pe_bwd": (c_i, [c_p] * 4 + [c_i] * 6 + [c_dp, c_p]),
    "mmvae_embed_ws_floats": (c_sz, [c_i] * 3),
    "mmvae_txt_layer_supported": (c_i, [c_i] * 5),
    "mmvae_txt_layer_lnws_floats": (c_sz, [c_i] * 3),
    "mmvae_txt_layer_fwd": (c_i, [c_p] * 7 + [c_i] * 7 + [c_p] * 3 + [c_i] + [c_p]),
    "mmvae_txt_layer_bwd": (c_i, [c_p] * 8 + [c_i] * 7 + [c_p]),
    "mmvae_attn_fwd": (c_i, [c_p] * 6 + [c_i] * 5 + [c_l] * 3 + [c_i, c_dp, c_p]),
    "mmvae_attn_bwd": (c_i, [c_p] * 8 + [c_i] * 5 + [c_l] * 3 + [c_dp, c_p]),
    "mmvae_layernorm_residual_fwd": (c_i, [c_p] * 7 + [c_i] * 3 + [c_dp, c_p]),
    "mmvae_layernorm_residual_bwd": (c_i, [c_p] * 9 + [c_i] * 3 + [c_dp, c_p]),
    "mmvae_layernorm_ws_floats": (c_sz, [c_i, c_i]),
    "mmvae_mean_over_time_fwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "mmvae_mean_over_time_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "mmvae_sum_over_time": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "mmvae_permute_mask_fwd": (c_i, [c_p] * 3 + [c_i] * 3 + [c_p]),
    "mmvae_permute_mask_bwd": (c_i, [c_p] * 3 + [c_i] * 3 + [c_p]),
    "mmvae_permute_mask_head_fwd": (c_i, [c_p] * 3 + [c_i] * 4 + [c_p]),
    "mmvae_permute_mask_head_bwd": (c_i, [c_p] * 3 + [c_i] * 4 + [c_p]),
    "mmvae_adam_amsgrad_flat": (c_i, [c_p] * 5 + [c_l] + [c_f] * 4 + [c_i, c_p, c_f, c_i, c_p]),
    "mmvae_adabelief_flat": (c_i, [c_p] * 4 + [c_l, c_f, ctypes.c_double, ctypes.c_double, c_f, c_i, c_p, c_f, c_i, c_p]),
    "mmvae_step_inc": (c_i, [c_p, c_p]),
    "mmvae_reduce_rows": (c_i, [c_p, c_p, c_i, c_l, c_l, c_i, c_p]),
    "mmvae_fill": (c_i, [c_p, c_l, c_f, c_p]),
    "mmvae_reduce_segments": (c_i, [c_p, c_p]),
    "mmvae_reduce_segments_lincomb": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "mmvae_ffn32_supported": (c_i, [c_i, c_i]),
    "mmvae_ffn32_bwd_parts": (c_i, [c_i, c_i]),
    "mmvae_ffn32_bwd_rowlen": (c_sz, [c_i]),
    "mmvae_ffn32_fwd": (c_i, [c_p] * 6 + [c_i, c_i, c_p, c_p]),
    "mmvae_ffn32_bwd": (c_i, [c_p] * 7 + [c_i, c_i, c_p, c_p]),
    "mmvae_proj32_ln_fwd": (c_i, [c_p] * 9 + [c_i, c_dp, c_p]),
    "mmvae_ffn32_wsplit_bytes": (c_sz, [c_i]),
    "mmvae_ffn32_rsplit_bytes": (c_sz, [c_i]),
    "mmvae_ffn32_prep_weights": (c_i, [c_p, c_p, c_p, c_i, c_p]),
    "mmvae_ffn32_prep_weights_many": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "mmvae_ffn32_fwd_b16": (c_i, [c_p] * 5 + [c_i, c_i, c_p, c_p]),
    "mmvae_ffn32_fwd_b16_ln": (c_i, [c_p] * 10 + [c_i, c_i, c_dp, c_dp, c_p]),
    "mmvae_ffn32_bwd_b16": (c_i, [c_p] * 8 + [c_i, c_i, c_p, c_p]),
    "mmvae_adam_fold_flat": (c_i, [c_p] * 5 + [c_l] + [c_f] * 4 + [c_p, c_f, c_i] + [c_p] * 4 + [c_i, c_i, c_i, c_p]),
    "mmvae_adam_fold_range": (c_i, [c_p] * 5 + [c_l, c_l, c_l, c_i] + [c_f] * 4 + [c_p, c_f, c_i] + [c_p] * 4
                              + [c_i, c_i, c_i, c_p]),
    "mmvae_rows_fan_fwd": (c_i, [c_p, c_p]),
    "mmvae_rows_fan_bwd": (c_i, [c_p, c_p]),
    "mmvae_normal_logratio_fwd": (c_i, [c_p] * 4 + [c_i, c_i, c_p]),
    "mmvae_normal_logratio_bwd": (c_i, [c_p] * 4 + [c_i, c_i, c_p]),
    "mmvae_expmul_fwd": (c_i, [c_p] * 3 + [c_i, c_p]),
    "mmvae_expmul_bwd": (c_i, [c_p] * 5 + [c_i, c_p]),
    "mmvae_moe_elbo_fwd": (c_i, [c_p, ctypes.POINTER(c_f), c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "mmvae_moe_elbo_bwd": (c_i, [c_p, c_p, ctypes.POINTER(c_f), c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "mmvae_moe_ksample_fwd": (c_i, [ctypes.POINTER(MoeKArgs), c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "mmvae_moe_ksample_bwd": (c_i, [ctypes.POINTER(MoeKBwdArgs), c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "mmvae_input_ring_pull": (c_i, [c_p, c_i, c_p, c_p, ctypes.c_size_t, ctypes.c_size_t, c_i, c_p]),
    "mmvae_gru_token_ids": (c_i, [c_p] * 3 + [c_i] * 3 + [c_p]),
    "mmvae_gru_forward": (c_i, [c_p] * 7 + [c_i] * 4 + [c_p]),
    "mmvae_gru_backward": (c_i, [c_p] * 6 + [c_i] * 3 + [c_p]),
    "mmvae_gru_cell0_fwd": (c_i, [c_p] * 7 + [c_i] * 3 + [c_p]),
    "mmvae_gru_cell0_bwd": (c_i, [c_p] * 5 + [c_i] * 2 + [c_p]),
    "mmvae_iwae_loss_out_doubles": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "mmvae_iwae_loss_fwd": (c_i, [c_p, ctypes.POINTER(DregRows), c_p, c_i, c_i, c_i, c_p]),
    "mmvae_iwae_loss_bwd": (c_i, [c_p, c_p, c_p, ctypes.POINTER(DregRows), c_p, c_i, c_i, c_i, c_p]),
    "mmvae_dreg_loss_fwd": (c_i, [c_p, ctypes.POINTER(DregRows), c_p, c_i, c_i, c_i, c_p]),
    "mmvae_dreg_loss_bwd": (c_i, [c_p, c_p, ctypes.POINTER(DregRows), c_p, c_i, c_i, c_i, c_p]),
    "mmvae_kl_laplace_normal_fwd": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "mmvae_kl_laplace_normal_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "mmvae_laplace_logratio_fwd": (c_i, [c_p] * 4 + [c_i, c_i, c_p]),
    "mmvae_laplace_logratio_bwd": (c_i, [c_p] * 4 + [c_i, c_i, c_p]),
    "mmvae_rand_laplace": (c_i, [c_p, c_l, c_p, c_p]),
    "mmvae_avgpool_fwd": (c_i, [c_p, c_p] + [c_i] * 4 + [c_p]),
    "mmvae_avgpool_bwd": (c_i, [c_p, c_p, c_p] + [c_i] * 4 + [c_p]),
    "mmvae_rc_tables": (c_i, [c_p, c_p] + [c_i] * 6 + [c_p]),
    "mmvae_rc_row_tile": (c_i, [c_i, c_i]),
    "mmvae_rc_conv_splits": (c_i, [c_i] * 4),
    "mmvae_rc_conv_ws_floats": (c_sz, [c_i] * 4),
    "mmvae_rc_launch": (c_i, [ctypes.POINTER(RcJob), c_i, c_p]),
    "mmvae_rc_bn_bwd_stats": (c_i, [c_p, ctypes.POINTER(RcStat), c_i, c_i, c_p]),
    "mmvae_rc_pool_bwd_stats": (c_i, [c_p, c_p, c_p, ctypes.POINTER(RcStat), c_i, c_i, c_i, c_p]),
    "mmvae_rc_stem_fwd": (c_i, [c_p] * 3 + [c_i] * 4 + [ctypes.POINTER(RcGeom)] + [c_p] * 9 + [c_f, c_f, c_i, c_p]),
    "mmvae_rc_maxpool_fwd": (c_i, [c_p] * 6 + [c_i] * 4 + [c_p]),
    "mmvae_rc_maxpool_bwd_stats": (c_i, [c_p] * 5 + [ctypes.POINTER(RcStat)] + [c_i] * 4 + [c_p]),
    "mmvae_rc_stem_wgrad_splits": (c_i, [c_i] * 4),
    "mmvae_rc_stem_wgrad_ws_floats": (c_sz, [c_i] * 4),
    "mmvae_rc_stem_wgrad": (c_i, [c_p] * 8 + [c_i] * 4 + [ctypes.POINTER(RcGeom), c_i, c_p]),
    "mmvae_rc_wgrad_splits": (c_i, [c_i] * 4),
    "mmvae_rc_wgrad_ws_floats": (c_sz, [c_i] * 4),
    "mmvae_rc_wgrad_tickets": (c_sz, [c_i] * 3),
    "mmvae_rc_blockout": (c_i, [c_p] * 8 + [c_i, c_p, c_l, c_i, c_p]),
    "mmvae_rc_bn_apply": (c_i, [c_p] * 5 + [c_l, c_i, c_p]),
    "mmvae_dropout_advance": (c_i, [c_p, c_u, c_p]),
    "mmvae_dropout_advance_many": (c_i, [c_p, c_i, c_p]),
    "mmvae_dropout_mask": (c_i, [c_dp, c_p, c_l, c_p]),
    "mmvae_dropout_act_fwd": (c_i, [c_p, c_p, c_l, c_i, c_dp, c_p]),
    "mmvae_dropout_act_bwd": (c_i, [c_p, c_p, c_p, c_l, c_i, c_dp, c_p]),
    "mmvae_head_bcast_dropout_fwd": (c_i, [c_p, c_p] + [c_i] * 4 + [c_dp, c_p]),
    "mmvae_head_bcast_dropout_bwd": (c_i, [c_p, c_p] + [c_i] * 4 + [c_dp, c_p]),
    "mmvae_conv_wgrad_layout": (c_i, [c_i] * 4 + [c_p] * 3),
    "mmvae_linear_bwd_weight_splits": (c_i, [c_i] * 3),
    "mmvae_linear_bwd_splits": (c_i, [c_i] * 3),
    "mmvae_layernorm_bwd_rows": (c_i, [c_i, c_i]),
    "mmvae_embed_bwd_rows": (c_i, [c_i] * 3),
}
ACC_DEFER = 2
MAX_SEGMENTS = 64


class RowPtrs(ctypes.Structure):
    _fields_ = [("p", c_p * 32)]


class GPtrs(ctypes.Structure):
    _fields_ = [("g", c_p * 4)]


INPUT_MAX_MODS = 8
INPUT_IMAGE_U8, INPUT_TEXT_TOKENS = 0, 1


class InputMod(ctypes.Structure):
    _fields_ = [("kind", c_i), ("src_off", c_sz), ("len_off", c_sz), ("dst", c_p), ("mask", c_p), ("n", c_l),
                ("B", c_i), ("T", c_i), ("V", c_i)]


WGRAD_BATCH_MAX = 8
DROPOUT_ADVANCE_MAX = 16


class WgradJob(ctypes.Structure):
    _fields_ = [("dy", c_p), ("x", c_p), ("dw", c_p), ("db", c_p), ("ws", c_p), ("M", c_i), ("N", c_i), ("K", c_i),
                ("ldx", c_l), ("x_act", c_i), ("accumulate", c_i)]


class TxtWgradJob(ctypes.Structure):      # mmvae_txt_wgrad_job_t
    _fields_ = [("dy", c_p), ("x", c_p), ("ws", c_p), ("M", c_i), ("N", c_i), ("K", c_i)]


TXT_WGRAD_MAX = 8


class TxtLayerW(ctypes.Structure):
    _fields_ = [(k, c_p) for k in ("in_w", "in_b", "out_w", "out_b", "l1_w", "l1_b", "l2_w", "l2_b", "n1_g", "n1_b",
                                   "n2_g", "n2_b", "n3_g", "n3_b", "x_in_w", "x_in_b", "x_out_w", "x_out_b")]


class TxtLayerSaved(ctypes.Structure):
    _fields_ = [(k, c_p) for k in ("qkv", "probs", "ao", "xhat1", "rstd1", "x1", "vproj", "vb", "xhat2", "rstd2", "x2",
                                   "h1", "g", "xhatf", "rstdf")]


class TxtLayerGrads(ctypes.Structure):
    _fields_ = [(k, c_p) for k in ("d_f", "d_h1", "d_ca", "d_v", "d_a", "d_qkv", "lnws")]


class TxtLayerDrop(ctypes.Structure):
    _fields_ = [(k, Dropout) for k in ("attn", "drop1", "xattn", "drop2", "ffn", "drop3")]


class ReduceSegments(ctypes.Structure):
    _fields_ = [("src", c_p * MAX_SEGMENTS), ("dst", c_p * MAX_SEGMENTS), ("rows", c_i * MAX_SEGMENTS),
                ("len", c_i * MAX_SEGMENTS), ("stride", c_i * MAX_SEGMENTS), ("blk0", c_i * MAX_SEGMENTS),
                ("next", c_i * MAX_SEGMENTS), ("n", c_i)]


def lib():
    """Load (once) and return the C-ABI library.  `import torch` has already loaded the HIP runtime
    (libamdhip64.so.7) this library links against, so both share one runtime and one set of streams."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with __graft_entry__.build() or "
                f"`make -C multimodal_vae_comparison_amd/csrc` (hipcc --offload-arch=gfx950). There is no fallback path.")
        # MMVAE_HIP_LIB: an alternative build of the SAME library (the -DMMVAE_TRACE debug build of tools/probe)
        L = ctypes.CDLL(os.environ.get("MMVAE_HIP_LIB", LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "HIP ops need tensors on the GPU (no CPU fallback)"
    return t.data_ptr()


def f32c(t):
    """contiguous fp32 view/copy of a CUDA tensor"""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


_ws = {}


def workspace(n_floats, device):
    """Caller-provided scratch for the *_ws_floats() contracts; grows monotonically per device.  Ops on one
    stream run in order, so consecutive ops may share it."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    t = _ws.get(key)
    if t is None or t.numel() < n_floats:
        t = torch.empty(max(int(n_floats), 1 << 20), dtype=torch.float32, device=device)
        _ws[key] = t
    return t


def reserve_workspace(n_floats, device):
    """Pre-size the workspace (call before hipGraph capture so capture never reallocates it)."""
    return workspace(n_floats, device)
