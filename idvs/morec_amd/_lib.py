"""ctypes binding of ``libmorec_hip.so`` (the C-ABI declared in ``include/morec_hip.h``).

There is NO fallback: if the shared library is missing or a call returns non-zero, this module raises.
The library is built in-tree by ``__graft_entry__.build()`` / ``make -C idvs/morec_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It has to be the one already mapped when libmorec_hip.so
# is loaded, otherwise the kernels would be launched through a second, uninitialised runtime ("no ROCm-capable device") while
# the streams and buffers they are handed belong to PyTorch's.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOREC_HIP_LIB: another build of the same ABI (A/B timing of kernel variants inside one GPU call)
LIB_PATH = os.environ.get("MOREC_HIP_LIB") or os.path.join(_HERE, "libmorec_hip.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
DACT_MUL = 3      # morec_gemm_desc.dact: multiply by dact_in (which holds act'(pre): aux_deriv outputs)


class MorecError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("in_dtype", C.c_int), ("out_dtype", C.c_int), ("act", C.c_int), ("dact", C.c_int),
                ("accumulate", C.c_int), ("split_k", C.c_int), ("alpha", C.c_float), ("aux_deriv", C.c_int)]


class AttnDesc(C.Structure):
    _fields_ = [("n_seq", C.c_int), ("T", C.c_int), ("n_heads", C.c_int), ("dh", C.c_int), ("causal", C.c_int),
                ("scale", C.c_float), ("mask_value", C.c_float), ("dtype", C.c_int), ("p_drop", C.c_float),
                ("seed", C.c_uint64), ("cu_seqlens", C.c_void_p), ("total_rows", C.c_int), ("spare_rows_max", C.c_int)]


class TransposeItem(C.Structure):      # morec_transpose_item
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld_src", C.c_int),
                ("ld_dst", C.c_int), ("tile0", C.c_int), ("reserved", C.c_int)]


class CeDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("S", C.c_int), ("D", C.c_int), ("Nc", C.c_int), ("col_offset", C.c_int),
                ("dtype", C.c_int), ("dE_fp32", C.c_int), ("ws_from_fwd", C.c_int)]


class StepParams(C.Structure):       # morec_step_params (64 bytes, lives on the DEVICE; this mirror is for host-side reads of a copy)
    _fields_ = [("step", C.c_int32), ("found_inf", C.c_int32), ("growth_tracker", C.c_int32), ("skipped", C.c_int32),
                ("loss_scale", C.c_float), ("inv_scale", C.c_float), ("bc1", C.c_float), ("bc2", C.c_float), ("apply", C.c_int32),
                ("pad0", C.c_int32), ("drop_seed", C.c_uint64), ("drop_seed_mixed", C.c_uint64), ("reserved", C.c_int32 * 2)]


class SwinAttnDesc(C.Structure):
    _fields_ = [("n_img", C.c_int), ("H", C.c_int), ("W", C.c_int), ("window", C.c_int), ("shift", C.c_int),
                ("heads", C.c_int), ("dh", C.c_int), ("scale", C.c_float), ("dtype", C.c_int)]


_P = C.c_void_p
_SIGS = {
    "morec_strerror": (C.c_char_p, [C.c_int]),
    "morec_version": (C.c_int, []),
    "morec_tuning_set": (C.c_int, [C.c_char_p, C.c_int]),
    "morec_det_scratch_reserve": (C.c_int, [C.c_size_t, C.c_void_p]),
    "morec_stream_wait_stream": (C.c_int, [_P, _P]),
    "morec_gemm_nt": (C.c_int, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P, _P, _P]),
    "morec_gemm_nt_colsum": (C.c_int, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "morec_gemm_colsum_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "morec_gemm_tn": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "morec_gemm_tn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "morec_mlp_dact_recompute_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "morec_mlp_dact_recompute_workspace_bytes": (C.c_size_t, [C.c_int]),
    "morec_mlp_dact_recompute": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_transpose": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_transpose_batch": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "morec_cast": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P]),
    "morec_split_bf16x3": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_act_bwd": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, C.c_int, _P]),
    "morec_scaled_sum": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_float, C.c_int, _P]),
    "morec_colsum": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_layernorm_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, C.c_float, _P, _P, _P, _P, C.c_int, C.c_int,
                                      C.c_int, C.c_float, C.c_uint64, C.c_float, C.c_uint64, _P, C.c_int, _P]),
    "morec_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_uint64, C.c_float, C.c_uint64, _P, _P, C.c_int, _P]),
    "morec_layernorm_fwd_res32": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, C.c_float, _P, _P, _P, _P, _P, C.c_int, C.c_int,
                                            C.c_int, C.c_float, C.c_uint64, C.c_float, C.c_uint64, _P]),
    "morec_layernorm_fwd_res32_pre": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                                C.c_float, C.c_uint64, _P]),
    "morec_layernorm_bwd_res32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float,
                                            C.c_uint64, C.c_float, C.c_uint64, _P]),
    "morec_pos_grad": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P]),
    "morec_attn_bwd": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P]),
    "morec_attn_bwd_dbias": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P, C.c_int, _P, _P, C.c_size_t, _P]),
    "morec_bert_embed_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_float, C.c_uint64, _P]),
    "morec_bert_embed_bwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "morec_gather_rows": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "morec_scatter_add_rows": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_indexed_rows_copy": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "morec_strided_rows_copy": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_inbatch_ce_workspace_bytes": (C.c_size_t, [C.POINTER(CeDesc)]),
    "morec_inbatch_ce_fwd": (C.c_int, [C.POINTER(CeDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "morec_inbatch_ce_bwd": (C.c_int, [C.POINTER(CeDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P,
                                       _P]),
    "morec_bce_fwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_bce_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_adamw": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                              C.c_int, C.c_float, _P]),
    "morec_dropout_seed_source": (C.c_int, [_P]),
    "morec_step_params_init": (C.c_int, [_P, C.c_float, C.c_int, _P]),
    "morec_grad_check_finite": (C.c_int, [_P, C.c_size_t, _P, _P]),
    "morec_step_decide": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _P]),
    "morec_adamw_sp": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "morec_eval_rank": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "morec_dropout_keep_mask": (C.c_int, [_P, C.c_size_t, C.c_float, C.c_uint64, _P]),
    "morec_probe": (C.c_int, [_P, _P]),
    "morec_swin_attn_fwd": (C.c_int, [C.POINTER(SwinAttnDesc), _P, _P, _P, _P]),
    "morec_swin_attn_bwd": (C.c_int, [C.POINTER(SwinAttnDesc), _P, _P, _P, _P, _P, _P, _P]),
    "morec_swin_attn_bwd_dbias": (C.c_int, [C.POINTER(SwinAttnDesc), _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "morec_swin_bias_expand": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "morec_swin_bias_reduce": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "morec_swin_patchify": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_swin_patchify_u8": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _P]),
    "morec_swin_merge": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_swin_pool_fwd": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_swin_pool_bwd": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "morec_bias_residual": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "morec_droppath_scale": (C.c_int, [_P, C.c_int, C.c_float, C.c_uint64, _P]),
    "morec_image_resize_u8": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "morec_comm_available": (C.c_int, []),
    "morec_comm_unique_id": (C.c_int, [_P]),
    "morec_comm_create": (C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_int]),
    "morec_comm_destroy": (C.c_int, [_P]),
    "morec_comm_last_error": (C.c_char_p, [_P]),
    "morec_comm_all_gather": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "morec_comm_reduce_scatter_f32": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "morec_comm_all_reduce_f32": (C.c_int, [_P, _P, C.c_size_t, _P]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises ``MorecError`` if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MorecError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "or `make -C idvs/morec_amd/csrc` -- there is no CPU/PyTorch fallback path")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = h
        # ONE parser for the deterministic switch: the Python decision (env_flag) is pushed into the library at load, so a value the
        # two sides would read differently ("true", "yes": C's atoi gives 0) cannot leave the engines and the kernels in different modes
        h.morec_tuning_set(b"deterministic", int(env_flag("MOREC_DETERMINISTIC")))
    return _lib


def env_flag(name: str) -> bool:
    """Boolean environment switch: unset / "" / "0" / "false" / "no" / "off" = off, anything else = on."""
    return os.environ.get(name, "0").strip().lower() not in ("", "0", "false", "no", "off")


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().morec_strerror(rc)
        raise MorecError(f"{what} failed: rc={rc} ({msg.decode() if msg else '?'})")
