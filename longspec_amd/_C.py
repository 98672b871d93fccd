"""ctypes binding of liblongspec_hip.so (the C ABI declared in include/longspec_hip.h).

The library is REQUIRED: there is no CPU or PyTorch fallback for any operator on
the hot path.  Import errors are raised loudly, with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# LONGSPEC_HIP_LIB: diagnostic builds of the same library (tools/build_variant.py, tools/ws_prof.py); the product path is the in-tree default
LIB_PATH = os.environ.get("LONGSPEC_HIP_LIB") or os.path.join(HERE, "_lib", "liblongspec_hip.so")

LS_F16, LS_BF16 = 0, 1
LS_NEW_NONE, LS_NEW_FLASH, LS_NEW_TARGET, LS_NEW_DRAFT = 0, 1, 2, 3


class AttnDesc(C.Structure):
    """Mirror of ``ls_attn_desc`` (include/longspec_hip.h) -- keep field order in sync."""
    _fields_ = [
        ("q", C.c_void_p), ("k_cache", C.c_void_p), ("v_cache", C.c_void_p),
        ("k_new", C.c_void_p), ("v_new", C.c_void_p), ("cache_seqlens", C.c_void_p),
        ("mask_bits", C.c_void_p), ("out", C.c_void_p), ("lse", C.c_void_p),
        ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p),
        ("b", C.c_int32), ("sq", C.c_int32), ("H", C.c_int32), ("Hkv", C.c_int32),
        ("dtype", C.c_int32), ("new_mode", C.c_int32), ("n_new", C.c_int32), ("n_new_cached", C.c_int32),
        ("mask_words", C.c_int32), ("scatter_new", C.c_int32), ("causal", C.c_int32), ("window_left", C.c_int32),
        ("n_app", C.c_int32), ("prescale_q", C.c_int32), ("kv_len_hint", C.c_int32), ("n_splits", C.c_int32),
        ("softmax_scale", C.c_float),
        ("q_stride_b", C.c_int64), ("q_stride_s", C.c_int64), ("q_stride_h", C.c_int64),
        ("kc_stride_b", C.c_int64), ("kc_stride_s", C.c_int64), ("kc_stride_h", C.c_int64),
        ("kn_stride_b", C.c_int64), ("kn_stride_s", C.c_int64), ("kn_stride_h", C.c_int64),
        ("out_stride_b", C.c_int64), ("out_stride_s", C.c_int64), ("out_stride_h", C.c_int64),
    ]


LS_EPI_NONE, LS_EPI_SILU_MUL, LS_EPI_QKV_ROPE = 0, 1, 2


class LinearDesc(C.Structure):
    """Mirror of ``ls_linear_desc`` (include/longspec_hip.h) -- keep field order in sync."""
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p * 3), ("bias", C.c_void_p * 3), ("y", C.c_void_p),
        ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p),
        ("M", C.c_int32), ("K", C.c_int32), ("n", C.c_int32 * 3), ("n_seg", C.c_int32),
        ("dtype", C.c_int32), ("epilogue", C.c_int32), ("n_splits", C.c_int32),
        ("ldx", C.c_int64), ("ldy", C.c_int64),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("norm_weight", C.c_void_p), ("ssq_in", C.c_void_p), ("ssq_out", C.c_void_p),
        ("ssq_parts", C.c_int32), ("norm_eps", C.c_float),
    ]


# every symbol include/longspec_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
SYMBOLS = {
    "ls_version": (C.c_int, []),
    "ls_last_error": (C.c_char_p, []),
    "ls_attn_workspace_bytes": (C.c_size_t, [C.POINTER(AttnDesc)]),
    "ls_attn_num_parts": (C.c_int, [C.POINTER(AttnDesc)]),
    "ls_attn_kernel_name": (C.c_char_p, [C.POINTER(AttnDesc)]),
    "ls_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), _P, C.c_size_t, _P]),
    "ls_attn_redo_count": (C.c_long, [_I]),
    "ls_attn_partial": (C.c_int, [C.POINTER(AttnDesc), _P, C.c_size_t, _P]),
    "ls_attn_reduce_local": (C.c_int, [C.POINTER(AttnDesc), _P, C.c_size_t, _P, _P, _P]),
    "ls_attn_finish": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _I, _L, _L, _P, C.c_size_t, _P]),
    "ls_linear_packed_bytes": (C.c_size_t, [_I, _I]),
    "ls_linear_pack_weight": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "ls_linear_pack_gate_up": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "ls_linear_pack_rope": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "ls_linear_workspace_bytes": (C.c_size_t, [C.POINTER(LinearDesc)]),
    "ls_linear_fwd": (C.c_int, [C.POINTER(LinearDesc), _P, C.c_size_t, _P]),
    "ls_linear_prefetch": (C.c_int, [C.POINTER(LinearDesc), _I, _P, C.c_size_t, _P]),
    "ls_topk_workspace_bytes": (C.c_size_t, [_I, _I, _I]),
    "ls_logprob_topk": (C.c_int, [_P, _I, _I, _L, _I, _P, _I, _P, _P, _P, C.c_size_t, _P]),
    "ls_argmax_rows": (C.c_int, [_P, _I, _I, _L, _I, _P, _P, C.c_size_t, _P]),
    "ls_topk_chunk": (C.c_int, []),
    "ls_topk_stage1": (C.c_int, [_P, _I, _I, _L, _I, _I, _I, _I, _P, _P]),
    "ls_topk_stage2": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "ls_lse_merge": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ls_pack_tree_mask": (C.c_int, [_P, _I, _I, _I, _P, _I, _P]),
    "ls_rmsnorm_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, C.c_float, _I, _P]),
    "ls_rope_cos_sin": (C.c_int, [_P, _P, C.c_float, _P, _P, _I, _I, _P]),
    "ls_rope_apply": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _L, _L, _I, _P]),
    "ls_tree_positions": (C.c_int, [_P, _P, _I, _I, _I, _P, _P]),
    "ls_tree_collapse": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _P]),
    "ls_tree_grow": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _P, _I, _P, _P, _I, _P]),
    "ls_tree_verify_inputs": (C.c_int, [_P, _L, _I, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P]),
    "ls_tree_verify_stochastic": (C.c_int, [_P, _P, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, C.c_float, _P, _I, _P, _P, _I, _P, _P,
                                            _P, _P]),
    "ls_tree_commit": (C.c_int, [_P, _P, _I, _I, _P, _L, _I, _I, _P, _I, _L, _P, _P, _P, _P, _I, _P, _I, _P, _P]),
    "ls_embed_rows": (C.c_int, [_P, _L, _I, _I, _P, _I, _P, _P]),
    "ls_pass_head": (C.c_int, [_P, _L, _I, _I, _P, _I, _P, _P, _I, _I, _P, C.c_float, _P, C.c_float, _P, _P, _P, _P, _P]),
    "ls_chain_commit": (C.c_int, [_P, _P, _I, _I, _P, _L, _I, _P, _P, _P, _P, _I, _L, _P, _P, _P]),
    "ls_xchg_create": (C.c_int, [_I, _I, C.c_size_t, C.POINTER(_P)]),
    "ls_xchg_handle": (C.c_int, [_P, _P]),
    "ls_xchg_connect": (C.c_int, [_P, _P]),
    "ls_xchg_all_gather": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "ls_xchg_status": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(_I)]),
    "ls_xchg_set_timeout": (C.c_int, [_P, C.c_double]),
    "ls_xchg_destroy": (C.c_int, [_P]),
    "ls_attn_reduce_push": (C.c_int, [C.POINTER(AttnDesc), _P, C.c_size_t, _P, _P]),
    "ls_attn_finish_xchg": (C.c_int, [C.POINTER(AttnDesc), _P, _P, C.c_size_t, _P]),
}


class LongSpecHipError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load the HIP extension; raise (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64: import torch FIRST so that this library binds to the
    # HIP runtime instance torch uses (device memory, streams and events are shared with it)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the MI355X HIP extension is required (no CPU/PyTorch fallback exists). "
            "Build it with `python -m longspec_amd.build` (needs hipcc; cross-compiles for gfx950 without a GPU).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)           # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ls_last_error()
        raise LongSpecHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
