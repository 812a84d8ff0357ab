"""ctypes binding of the C-ABI declared in include/mxvl.h (libmxvl.so).

This is the only place Python touches the native library.  Tensors cross the boundary as raw
device pointers + element strides; the stream is torch's current HIP stream (the reference
enqueues on at::cuda::getCurrentCUDAStream(), selective_scan.cpp:232-233).  Loading fails loudly:
there is no fallback implementation anywhere in the package.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_int32, c_int64, c_uint32, c_void_p

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libmxvl.so")
ABI_VERSION = 12

MXVL_F32, MXVL_BF16, MXVL_F16 = 0, 1, 2
SCAN_DELTA_SOFTPLUS = 1
SCAN_OUT_F32 = 2          # out / dout fp32 whatever io_dtype is (the oflex i16o32 mode)
SCAN_FOLD_BATCH = 4       # the batch folded into the sequence (short rows; ask mxvl_scan_fold_ok)

STATUS = {
    0: "MXVL_OK", -1: "MXVL_ERR_NULL", -2: "MXVL_ERR_DTYPE", -3: "MXVL_ERR_SHAPE", -4: "MXVL_ERR_DSTATE",
    -5: "MXVL_ERR_STRIDE", -6: "MXVL_ERR_LAUNCH", -7: "MXVL_ERR_UNSUPPORTED", -8: "MXVL_ERR_CHECKPOINT",
}

# every symbol include/mxvl.h declares (tests/test_abi.py checks the .so exports all of them)
SYMBOLS = [
    "mxvl_abi_version", "mxvl_scan_chunk_len", "mxvl_scan_n_chunks", "mxvl_scan_fold_ok", "mxvl_scan_fold_slots", "mxvl_scan_fwd", "mxvl_scan_bwd",
    "mxvl_scan_bwd_workspace_bytes",
    "mxvl_conv1d_fwd", "mxvl_conv1d_bwd", "mxvl_conv1d_update", "mxvl_state_update",
    "mxvl_last_hip_error", "mxvl_set_scan_variant", "mxvl_last_scan_kernel", "mxvl_decode_gemv", "mxvl_decode_attn",
    "mxvl_decode_cross_attn", "mxvl_decode_prologue", "mxvl_decode_rmsnorm",
    "mxvl_cross_scan", "mxvl_cross_merge",
    "mxvl_add_layernorm_fwd", "mxvl_add_layernorm_bwd", "mxvl_add_layernorm_partials", "mxvl_swiglu_fwd", "mxvl_swiglu_bwd",
    "mxvl_swiglu_partials", "mxvl_swiglu_bwd_colsum", "mxvl_gemm_swiglu_fwd", "mxvl_gemm_swiglu_bwd", "mxvl_gemm_swiglu_bwd_partials", "mxvl_set_decode_gemm_wide", "mxvl_decode_gemm_plan", "mxvl_gemm_nt", "mxvl_gemm_tn", "mxvl_colsum", "mxvl_colsum_partials", "mxvl_row_gather", "mxvl_patch_loss", "mxvl_patch_cols", "mxvl_beam_workspace_bytes",
    "mxvl_dwconv2d_fwd", "mxvl_dwconv2d_bwd", "mxvl_beam_step", "mxvl_dir_gather", "mxvl_dir_merge",
    "mxvl_resample_ksize", "mxvl_resample_coeffs", "mxvl_image_preprocess", "mxvl_attn_fwd", "mxvl_attn_bwd", "mxvl_clip_loss",
    "mxvl_rope", "mxvl_rmsnorm_train_fwd", "mxvl_rmsnorm_train_bwd", "mxvl_silu_mul",
    "mxvl_window_cols", "mxvl_mamba_inner_fwd", "mxvl_mamba_inner_bwd", "mxvl_mamba_inner_workspace_bytes", "mxvl_mamba_inner_bwd_workspace_bytes",
]


class ScanDesc(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32), ("dim", c_int32), ("seqlen", c_int32), ("dstate", c_int32), ("n_groups", c_int32),
        ("io_dtype", c_int32), ("flags", c_uint32), ("delta_group_ratio", c_int32),
        ("u_bs", c_int64), ("u_ds", c_int64), ("delta_bs", c_int64), ("delta_ds", c_int64),
        ("z_bs", c_int64), ("z_ds", c_int64), ("out_bs", c_int64), ("out_ds", c_int64),
        ("B_bs", c_int64), ("B_gs", c_int64), ("B_ns", c_int64),
        ("C_bs", c_int64), ("C_gs", c_int64), ("C_ns", c_int64),
        ("A_ds", c_int64), ("A_ns", c_int64),
        ("u", c_void_p), ("delta", c_void_p), ("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
        ("D", c_void_p), ("delta_bias", c_void_p), ("z", c_void_p),
        ("out", c_void_p), ("last_state", c_void_p), ("ckpt", c_void_p),
    ]


class ScanBwdDesc(ctypes.Structure):
    _fields_ = [
        ("fwd", ScanDesc),
        ("dout_bs", c_int64), ("dout_ds", c_int64), ("du_bs", c_int64), ("du_ds", c_int64),
        ("ddelta_bs", c_int64), ("ddelta_ds", c_int64), ("dz_bs", c_int64), ("dz_ds", c_int64),
        ("dB_bs", c_int64), ("dB_gs", c_int64), ("dB_ns", c_int64),
        ("dC_bs", c_int64), ("dC_gs", c_int64), ("dC_ns", c_int64),
        ("dout", c_void_p), ("du", c_void_p), ("ddelta", c_void_p), ("dz", c_void_p),
        ("dA", c_void_p), ("dB", c_void_p), ("dC", c_void_p), ("dD", c_void_p), ("ddelta_bias", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_int64),
    ]


class MambaInnerDesc(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32), ("dim", c_int32), ("seqlen", c_int32), ("dstate", c_int32), ("dt_rank", c_int32), ("width", c_int32),
        ("d_model", c_int32), ("io_dtype", c_int32), ("flags", c_uint32), ("reserved0", c_int32),
        ("xz", c_void_p), ("conv_weight", c_void_p), ("conv_bias", c_void_p), ("x_proj_weight", c_void_p), ("dt_proj_weight", c_void_p),
        ("out_proj_weight", c_void_p), ("out_proj_bias", c_void_p), ("A", c_void_p), ("D", c_void_p), ("delta_bias", c_void_p),
        ("out", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_int64),
    ]


class MambaInnerBwdDesc(ctypes.Structure):
    _fields_ = [
        ("fwd", MambaInnerDesc),
        ("dout", c_void_p), ("dxz", c_void_p), ("dconv_weight", c_void_p), ("dconv_bias", c_void_p),
        ("dx_proj_weight", c_void_p), ("ddt_proj_weight", c_void_p), ("dout_proj_weight", c_void_p), ("dout_proj_bias", c_void_p),
        ("dA", c_void_p), ("dD", c_void_p), ("ddelta_bias", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_int64),
    ]


class AttnDesc(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32), ("n_heads", c_int32), ("n_kv_heads", c_int32), ("seqlen_q", c_int32), ("seqlen_k", c_int32),
        ("head_dim", c_int32), ("io_dtype", c_int32), ("mask_mode", c_int32), ("cluster", c_int32), ("scale", ctypes.c_float),
        ("q_bs", c_int64), ("q_hs", c_int64), ("q_ts", c_int64), ("k_bs", c_int64), ("k_hs", c_int64), ("k_ts", c_int64),
        ("v_bs", c_int64), ("v_hs", c_int64), ("v_ts", c_int64), ("o_bs", c_int64), ("o_hs", c_int64), ("o_ts", c_int64),
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("out", c_void_p), ("lse", c_void_p), ("key_mask", c_void_p),
        ("bias", c_void_p), ("dropout_p", ctypes.c_float), ("dropout_seed", c_uint32),
    ]


class AttnBwdDesc(ctypes.Structure):
    _fields_ = [
        ("fwd", AttnDesc),
        ("dout_bs", c_int64), ("dout_hs", c_int64), ("dout_ts", c_int64), ("dq_bs", c_int64), ("dq_hs", c_int64), ("dq_ts", c_int64),
        ("dk_bs", c_int64), ("dk_hs", c_int64), ("dk_ts", c_int64), ("dv_bs", c_int64), ("dv_hs", c_int64), ("dv_ts", c_int64),
        ("dout", c_void_p), ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p), ("delta", c_void_p),
    ]


class Conv1dDesc(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32), ("dim", c_int32), ("seqlen", c_int32), ("width", c_int32),
        ("io_dtype", c_int32), ("silu", c_int32),
        ("x_bs", c_int64), ("x_ds", c_int64), ("y_bs", c_int64), ("y_ds", c_int64),
        ("x", c_void_p), ("weight", c_void_p), ("bias", c_void_p), ("y", c_void_p),
    ]


class Conv1dBwdDesc(ctypes.Structure):
    _fields_ = [
        ("fwd", Conv1dDesc),
        ("dy_bs", c_int64), ("dy_ds", c_int64), ("dx_bs", c_int64), ("dx_ds", c_int64),
        ("dy", c_void_p), ("dx", c_void_p), ("dweight", c_void_p), ("dbias", c_void_p),
    ]


class GemvDesc(ctypes.Structure):
    _fields_ = [
        ("rows", c_int32), ("K", c_int32), ("N", c_int32), ("swiglu", c_int32), ("out_f32", c_int32),
        ("eps", ctypes.c_float),
        ("x", c_void_p), ("norm_weight", c_void_p), ("W", c_void_p), ("W2", c_void_p), ("bias", c_void_p),
        ("residual", c_void_p), ("y", c_void_p), ("split_acc", c_void_p), ("k_splits", c_int32), ("dtype", c_int32),
        ("norm_gain_scale", ctypes.c_float),
    ]


class RmsNormDesc(ctypes.Structure):
    _fields_ = [("rows", c_int32), ("K", c_int32), ("eps", ctypes.c_float), ("x", c_void_p), ("weight", c_void_p), ("y", c_void_p),
                ("acc", c_void_p), ("residual", c_void_p), ("x_out", c_void_p), ("dtype", c_int32), ("acc_splits", c_int32)]


class RopeDesc(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32), ("seqlen", c_int32), ("n_q_heads", c_int32), ("n_k_heads", c_int32), ("head_dim", c_int32),
        ("io_dtype", c_int32), ("cs_dtype", c_int32), ("backward", c_int32),
        ("q_bs", c_int64), ("q_ts", c_int64), ("q_hs", c_int64), ("k_bs", c_int64), ("k_ts", c_int64), ("k_hs", c_int64),
        ("qo_bs", c_int64), ("qo_ts", c_int64), ("qo_hs", c_int64), ("ko_bs", c_int64), ("ko_ts", c_int64), ("ko_hs", c_int64),
        ("cs_bs", c_int64), ("cs_ts", c_int64),
        ("q", c_void_p), ("k", c_void_p), ("cos", c_void_p), ("sin", c_void_p), ("q_out", c_void_p), ("k_out", c_void_p),
    ]


class RmsTrainDesc(ctypes.Structure):
    _fields_ = [("rows", c_int32), ("cols", c_int32), ("x_dtype", c_int32), ("w_dtype", c_int32), ("y_dtype", c_int32),
                ("eps", ctypes.c_float), ("x", c_void_p), ("weight", c_void_p), ("grad", c_void_p), ("y", c_void_p), ("rstd", c_void_p)]


class DecodeAttnDesc(ctypes.Structure):
    _fields_ = [
        ("rows", c_int32), ("n_heads", c_int32), ("n_kv_heads", c_int32), ("head_dim", c_int32), ("max_len", c_int32),
        ("scale", ctypes.c_float),
        ("qkv", c_void_p), ("cos", c_void_p), ("sin", c_void_p), ("k_cache", c_void_p), ("v_cache", c_void_p),
        ("slot_table", c_void_p), ("pos", c_void_p), ("mask", c_void_p), ("out", c_void_p), ("q_rope", c_void_p),
        ("beams", c_int32), ("dtype", c_int32),
    ]


GATE_TANH, GATE_WARM_TANH = 1, 2


class DecodeCrossAttnDesc(ctypes.Structure):
    _fields_ = [
        ("rows", c_int32), ("n_heads", c_int32), ("n_kv_heads", c_int32), ("head_dim", c_int32), ("n_keys", c_int32),
        ("kv_rows_div", c_int32), ("gate_flags", c_int32), ("scale", ctypes.c_float),
        ("q_rope", c_void_p), ("k", c_void_p), ("v", c_void_p), ("key_mask", c_void_p), ("row_on", c_void_p),
        ("text_state", c_void_p), ("gate_weight", c_void_p), ("gate_bias", c_void_p), ("warm_up_gate", c_void_p),
        ("out", c_void_p), ("dtype", c_int32), ("reserved0", c_int32),
    ]


class DecodePrologueDesc(ctypes.Structure):
    _fields_ = [
        ("rows", c_int32), ("hidden", c_int32), ("max_len", c_int32), ("head_dim", c_int32), ("prompt_len", c_int32),
        ("table_len", c_int32),
        ("tok", c_void_p), ("beam_src", c_void_p), ("cur", c_void_p), ("n_real", c_void_p), ("embed", c_void_p),
        ("cos_table", c_void_p), ("sin_table", c_void_p), ("slot_table", c_void_p), ("mask", c_void_p), ("x", c_void_p),
        ("cos", c_void_p), ("sin", c_void_p), ("pos", c_void_p),
    ]


class AddLnDesc(ctypes.Structure):
    _fields_ = [
        ("rows", c_int32), ("cols", c_int32), ("res_dtype", c_int32), ("branch_dtype", c_int32), ("out_dtype", c_int32),
        ("eps", ctypes.c_float),
        ("x", c_void_p), ("branch", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
        ("h", c_void_p), ("n", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
    ]


class GemmSwigluDesc(ctypes.Structure):
    _fields_ = [
        ("M", c_int32), ("K", c_int32), ("H", c_int32), ("io_dtype", c_int32), ("bias_dtype", c_int32),
        ("x_rs", c_int64), ("w_rs", c_int64), ("ab_rs", c_int64), ("h_rs", c_int64),
        ("x", c_void_p), ("weight", c_void_p), ("bias", c_void_p), ("ab", c_void_p), ("h", c_void_p),
    ]


class GemmNtDesc(ctypes.Structure):
    _fields_ = [
        ("M", c_int32), ("K", c_int32), ("N", c_int32), ("io_dtype", c_int32), ("bias_dtype", c_int32), ("reserved0", c_int32),
        ("a_rs", c_int64), ("b_rs", c_int64), ("c_rs", c_int64),
        ("a", c_void_p), ("b", c_void_p), ("bias", c_void_p), ("c", c_void_p),
    ]


class GemmTnDesc(ctypes.Structure):
    _fields_ = [
        ("M", c_int32), ("N", c_int32), ("K", c_int32), ("io_dtype", c_int32), ("accumulate", c_int32), ("slices_per_xcd", c_int32),
        ("a_rs", c_int64), ("b_rs", c_int64), ("c_rs", c_int64),
        ("a", c_void_p), ("b", c_void_p), ("c", c_void_p),
    ]


class GemmSwigluBwdDesc(ctypes.Structure):
    _fields_ = [
        ("M", c_int32), ("K", c_int32), ("H", c_int32), ("io_dtype", c_int32),
        ("dy_rs", c_int64), ("w_rs", c_int64), ("ab_rs", c_int64), ("dab_rs", c_int64),
        ("dy", c_void_p), ("w3t", c_void_p), ("ab", c_void_p), ("dab", c_void_p), ("partial", c_void_p),
    ]


class AddLnBwdDesc(ctypes.Structure):
    _fields_ = [
        ("rows", c_int32), ("cols", c_int32), ("res_dtype", c_int32), ("branch_dtype", c_int32), ("out_dtype", c_int32),
        ("n_partials", c_int32),
        ("dn", c_void_p), ("dh", c_void_p), ("h", c_void_p), ("gamma", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
        ("dx", c_void_p), ("dbranch", c_void_p), ("partial_dgamma", c_void_p), ("partial_dbeta", c_void_p),
        ("partial_dbranch", c_void_p),
    ]


class DirPermDesc(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32), ("dim", c_int32), ("seqlen", c_int32), ("padded_len", c_int32), ("n_dirs", c_int32), ("io_dtype", c_int32),
        ("rows_bs", c_int64), ("rows_ds", c_int64), ("stacked_bs", c_int64), ("stacked_ks", c_int64), ("stacked_ds", c_int64),
        ("index", c_void_p), ("rows", c_void_p), ("stacked", c_void_p),
        ("gate", c_void_p), ("pre", c_void_p), ("dgate", c_void_p),
        ("gate_bs", c_int64), ("gate_ds", c_int64), ("pre_bs", c_int64), ("pre_ds", c_int64), ("dgate_bs", c_int64), ("dgate_ds", c_int64),
        ("gate_scale", ctypes.c_float), ("reserved0", c_int32),
    ]


class BeamDesc(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32), ("beams", c_int32), ("vocab", c_int32), ("max_new", c_int32), ("min_new", c_int32), ("n_eos", c_int32),
        ("early_stopping", c_int32), ("keep", c_int32),
        ("repetition_penalty", ctypes.c_float), ("reserved0", c_int32),
        ("logits", c_void_p), ("run_seq", c_void_p), ("fin_seq", c_void_p), ("run_score", c_void_p), ("fin_score", c_void_p),
        ("fin_done", c_void_p), ("heur_open", c_void_p), ("cur", c_void_p), ("eos", c_void_p), ("len_tab", c_void_p),
        ("hyp_tab", c_void_p), ("tok", c_void_p), ("beam_src", c_void_p), ("unfinished", c_void_p), ("scratch", c_void_p),
        ("unfinished_log", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_int64),
    ]


class ImageDesc(ctypes.Structure):
    _fields_ = [
        ("in_h", c_int32), ("in_w", c_int32), ("out_h", c_int32), ("out_w", c_int32),
        ("ksize_h", c_int32), ("ksize_v", c_int32), ("out_dtype", c_int32), ("reserved0", c_int32),
        ("src", c_void_p), ("bounds_h", c_void_p), ("kk_h", c_void_p), ("bounds_v", c_void_p), ("kk_v", c_void_p),
        ("lut", c_void_p), ("tmp", c_void_p), ("out", c_void_p),
    ]


_lib = None


def load() -> ctypes.CDLL:
    """dlopen libmxvl.so; raises (never falls back) when it is absent or its ABI is stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m medical_image_analysis_amd.build` "
            "(hipcc, gfx950). medical_image_analysis_amd has no CPU / PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.mxvl_abi_version.restype = c_int
    if lib.mxvl_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libmxvl.so ABI {lib.mxvl_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    lib.mxvl_last_scan_kernel.restype = ctypes.c_char_p
    lib.mxvl_scan_bwd_workspace_bytes.restype = c_int64
    for name in ("mxvl_scan_fwd", "mxvl_scan_bwd", "mxvl_conv1d_fwd", "mxvl_conv1d_bwd", "mxvl_decode_gemv", "mxvl_decode_attn",
                 "mxvl_decode_cross_attn", "mxvl_attn_fwd", "mxvl_attn_bwd", "mxvl_gemm_swiglu_fwd", "mxvl_gemm_swiglu_bwd", "mxvl_gemm_nt", "mxvl_gemm_tn"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = [c_void_p, c_void_p]
    for name in ("mxvl_mamba_inner_fwd", "mxvl_mamba_inner_bwd"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = [c_void_p, c_void_p]
    for name in ("mxvl_mamba_inner_workspace_bytes", "mxvl_mamba_inner_bwd_workspace_bytes"):
        getattr(lib, name).restype = c_int64
        getattr(lib, name).argtypes = [c_void_p]
    lib.mxvl_conv1d_update.restype = c_int
    lib.mxvl_conv1d_update.argtypes = [c_void_p] * 5 + [c_int] * 5 + [c_void_p]
    lib.mxvl_state_update.restype = c_int
    lib.mxvl_state_update.argtypes = [c_void_p] * 10 + [c_int] * 5 + [c_void_p]
    for name in ("mxvl_dir_gather", "mxvl_dir_merge"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = [c_void_p, c_void_p]
    lib.mxvl_beam_step.restype = c_int
    lib.mxvl_beam_step.argtypes = [c_void_p, c_void_p]
    for name in ("mxvl_decode_prologue", "mxvl_decode_rmsnorm"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = [c_void_p, c_void_p]
    for name in ("mxvl_rope", "mxvl_rmsnorm_train_fwd", "mxvl_rmsnorm_train_bwd"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = [c_void_p, c_void_p]
    lib.mxvl_silu_mul.restype = c_int
    lib.mxvl_silu_mul.argtypes = [c_void_p] * 5 + [c_int64, c_int, c_void_p]
    for name in ("mxvl_add_layernorm_fwd", "mxvl_add_layernorm_bwd"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = [c_void_p, c_void_p]
    lib.mxvl_add_layernorm_partials.restype = c_int
    lib.mxvl_add_layernorm_partials.argtypes = [c_int]
    lib.mxvl_gemm_swiglu_bwd_partials.restype = c_int
    lib.mxvl_gemm_swiglu_bwd_partials.argtypes = [c_int]
    lib.mxvl_set_decode_gemm_wide.restype = c_int
    lib.mxvl_set_decode_gemm_wide.argtypes = [c_int]
    lib.mxvl_decode_gemm_plan.restype = c_int
    lib.mxvl_decode_gemm_plan.argtypes = [ctypes.POINTER(GemvDesc), ctypes.POINTER(c_int32)]
    lib.mxvl_swiglu_fwd.restype = c_int
    lib.mxvl_swiglu_fwd.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.mxvl_swiglu_bwd.restype = c_int
    lib.mxvl_swiglu_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.mxvl_swiglu_partials.restype = c_int
    lib.mxvl_swiglu_partials.argtypes = [c_int, c_int]
    lib.mxvl_colsum_partials.restype = c_int
    lib.mxvl_colsum_partials.argtypes = [c_int, c_int]
    lib.mxvl_colsum.restype = c_int
    lib.mxvl_colsum.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p]
    lib.mxvl_swiglu_bwd_colsum.restype = c_int
    lib.mxvl_swiglu_bwd_colsum.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.mxvl_dwconv2d_fwd.restype = c_int
    lib.mxvl_dwconv2d_fwd.argtypes = [c_void_p] * 4 + [c_int] * 7 + [c_void_p]
    lib.mxvl_dwconv2d_bwd.restype = c_int
    lib.mxvl_dwconv2d_bwd.argtypes = [c_void_p] * 7 + [c_int] * 7 + [c_void_p]
    for name in ("mxvl_cross_scan", "mxvl_cross_merge"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]
    lib.mxvl_resample_ksize.restype = c_int
    lib.mxvl_resample_ksize.argtypes = [c_int, c_int, c_int]
    lib.mxvl_resample_coeffs.restype = c_int
    lib.mxvl_resample_coeffs.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p]
    lib.mxvl_image_preprocess.restype = c_int
    lib.mxvl_image_preprocess.argtypes = [c_void_p, c_void_p]
    lib.mxvl_clip_loss.restype = c_int
    lib.mxvl_clip_loss.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.mxvl_row_gather.restype = c_int
    lib.mxvl_row_gather.argtypes = [c_void_p] * 5 + [c_int] * 4 + [c_int64] * 2 + [c_int] * 2 + [c_void_p]
    lib.mxvl_patch_loss.restype = c_int
    lib.mxvl_patch_loss.argtypes = [c_void_p] * 5 + [c_int] * 6 + [c_void_p]
    lib.mxvl_beam_workspace_bytes.restype = c_int64
    lib.mxvl_beam_workspace_bytes.argtypes = [c_int] * 3
    lib.mxvl_window_cols.restype = c_int
    lib.mxvl_window_cols.argtypes = [c_void_p] * 3 + [c_int] * 8 + [c_void_p]
    lib.mxvl_patch_cols.restype = c_int
    lib.mxvl_patch_cols.argtypes = [c_void_p] * 2 + [c_int] * 7 + [c_void_p]
    lib.mxvl_scan_chunk_len.restype = c_int
    lib.mxvl_scan_n_chunks.restype = c_int
    lib.mxvl_scan_fold_ok.restype = c_int
    lib.mxvl_scan_fold_ok.argtypes = [c_int, c_int, c_int]
    lib.mxvl_scan_fold_slots.restype = c_int
    lib.mxvl_scan_fold_slots.argtypes = [c_int, c_int, c_int, c_int]
    lib.mxvl_set_scan_variant.argtypes = [c_int]
    _lib = lib
    return lib


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return MXVL_F32
    if dt == torch.bfloat16:
        return MXVL_BF16
    if dt == torch.float16:
        return MXVL_F16
    raise RuntimeError(f"mxvl: io dtype must be float32/bfloat16/float16, got {dt}")  # selective_scan.cpp:167


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(*tensors) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mxvl: expected a HIP device tensor (no CPU path exists); got a CPU tensor")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("mxvl: tensors live on different devices")
    return dev


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        hip = lib.mxvl_last_hip_error()
        raise RuntimeError(f"{what} failed: {STATUS.get(rc, rc)} (hipError {hip})")
