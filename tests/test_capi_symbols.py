"""The C-ABI library builds, loads and exports every symbol include/longspec_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "longspec_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ls_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from longspec_amd.build import build
    from longspec_amd import _C
    lib_path = build(verbose=False)
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in longspec_hip.h but not exported"
        assert n in _C.SYMBOLS, f"{n} has no ctypes prototype in longspec_amd/_C.py"
    assert sorted(_C.SYMBOLS) == names
    assert _C.load().ls_version() >= 100


def test_attn_desc_layout_matches_header():
    """Field order of the ctypes mirror == field order of ls_attn_desc."""
    from longspec_amd import _C
    txt = open(os.path.join(ROOT, "include", "longspec_hip.h")).read()
    body = txt[txt.index("typedef struct ls_attn_desc {"):txt.index("} ls_attn_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        decl = stmt.split(None, 1)
        names = stmt.replace("*", " ").split()
        # "const void* q" / "int32_t b, sq, H, Hkv" / "int64_t a, b, c"
        tail = stmt
        for kw in ("const", "void", "uint32_t", "int32_t", "int64_t", "float", "*"):
            tail = tail.replace(kw, " ")
        fields += [f.strip() for f in tail.split(",") if f.strip()]
    assert fields == [f[0] for f in _C.AttnDesc._fields_]
    assert ctypes.sizeof(_C.AttnDesc) == 11 * 8 + 16 * 4 + 4 + 4 + 12 * 8


def test_linear_desc_layout_matches_header():
    """Field order and size of the ctypes mirror == ls_linear_desc."""
    from longspec_amd import _C
    txt = open(os.path.join(ROOT, "include", "longspec_hip.h")).read()
    body = txt[txt.index("typedef struct ls_linear_desc {"):txt.index("} ls_linear_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for stmt in body.split("{", 1)[1].split(";"):
        tail = stmt.strip()
        if not tail:
            continue
        tail = re.sub(r"\[\d+\]", "", tail)
        for kw in ("const", "void", "int32_t", "int64_t", "float", "*"):
            tail = tail.replace(kw, " ")
        fields += [f.strip() for f in tail.split(",") if f.strip()]
    assert fields == [f[0] for f in _C.LinearDesc._fields_]
    assert ctypes.sizeof(_C.LinearDesc) == (1 + 3 + 3 + 1 + 2) * 8 + (2 + 3 + 4) * 4 + 4 + 2 * 8 + 2 * 8 + 2 * 8 + 3 * 8 + 2 * 4


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from longspec_amd import ops
    x = torch.zeros(2, 8, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.rmsnorm(x, torch.ones(8, dtype=torch.float16), 1e-5)
