"""ctypes wrapper of oracle/_build/liboracle.so (built by oracle/Makefile).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = C.CDLL(LIB)
        for fn in (_lib.oracle_verify_attention_f16, _lib.oracle_verify_attention_bf16):
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    return _lib


def verify_attention(q, k_new, v_new, k_cache, v_cache, L: int, tree_mask, last_layer: bool, scale=1.0 / (128 ** 0.5)):
    """q [1,R,H,128] fp16 or bf16 etc. (CPU, contiguous), caches [1,S,Hkv,128] modified in place. Returns [1,R,H,128] in q's dtype."""
    lib = load()
    _, R, H, D = q.shape
    Hkv = k_new.shape[2]
    out = torch.empty_like(q)
    tm = tree_mask.to(torch.int64).contiguous()
    for t in (q, k_new, v_new, k_cache, v_cache):
        assert t.is_contiguous() and t.dtype == q.dtype and q.dtype in (torch.float16, torch.bfloat16) and not t.is_cuda
    fn = lib.oracle_verify_attention_f16 if q.dtype == torch.float16 else lib.oracle_verify_attention_bf16
    rc = fn(q.data_ptr(), k_new.data_ptr(), v_new.data_ptr(), k_cache.data_ptr(),
            v_cache.data_ptr(), int(L), tm.data_ptr(), R, H, Hkv, int(last_layer), float(scale), out.data_ptr())
    assert rc == 0
    return out
