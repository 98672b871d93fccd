#!/usr/bin/env python3
"""Build a diagnostic variant of the library:
    python tools/build_variant.py NAME [--src gemm.hip ...] -DFOO=1 ...   ->   longspec_amd/_lib/liblongspec_hip_NAME.so
(default source: attn.hip).  Load it with LONGSPEC_HIP_LIB=<path>."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longspec_amd import build as b
name, rest = sys.argv[1], sys.argv[2:]
srcs, flags = [], []
it = iter(rest)
for a in it:
    if a == "--src":
        srcs.append(next(it))
    else:
        flags.append(a)
print(b.build_variant(name, flags, force=True, sources=tuple(srcs) or ("attn.hip",)))
