#!/usr/bin/env python3
"""Build a diagnostic variant of the library: python tools/build_variant.py NAME -DFOO=1 ... -> _lib/liblongspec_hip_NAME.so"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longspec_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
objs = []
for src in b.SOURCES:
    obj = os.path.join(b.LIBDIR, src.replace(".hip", ".o"))
    if src in ("attn.hip", "xgmi.hip"):
        obj = os.path.join(b.LIBDIR, src.replace(".hip", f"_{name}.o"))
        subprocess.check_call([b._hipcc()] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj])
    objs.append(obj)
out = os.path.join(b.LIBDIR, f"liblongspec_hip_{name}.so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print(out)
