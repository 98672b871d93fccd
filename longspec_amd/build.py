"""Build liblongspec_hip.so (gfx950 only) in-tree with hipcc.

    python -m longspec_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting ``longspec_amd/_lib/liblongspec_hip.so``
travels to the GPU box with the repository snapshot (it is git-ignored, not
gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "liblongspec_hip.so")
SOURCES = ["attn.hip", "gemm.hip", "misc.hip", "topk.hip", "tree.hip", "xgmi.hip"]
HEADERS = [os.path.join(CSRC, "ls_common.h"), os.path.join(os.path.dirname(HERE), "include", "longspec_hip.h")]
# -ffp-contract=off: the reference-order roundings (fp16 product, fp16 sum) must not be fused into FMAs
# -Wno-inline-asm: the LDS-DMA helper names m0 in its clobber list on purpose.  -Wno-division-by-zero: the HOST pass folds
# __builtin_amdgcn_kernarg_segment_ptr() to null and then flags `m / p.sq` in device-only code it never emits.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=off", "-Wno-inline-asm",
         "-Wno-division-by-zero"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _write_usage(remarks: str, path: str) -> None:
    """Per-kernel register / scratch figures from the compiler's -Rpass-analysis=kernel-resource-usage remarks, kept next to the
    object (``_lib/<src>.usage.json``).  tests/test_build_resources.py reads them: a streaming kernel that starts spilling (one
    innocent-looking edit of the split arithmetic cost attn_partial_ws_kernel 512 registers of scratch and a factor 3.4 in
    round 3) is a build failure, not something to find with a profiler."""
    import json
    import re
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        usage = obj.replace(".o", ".usage.json")
        if force or _stale(obj, [sp] + HEADERS) or not os.path.exists(usage):
            cmd = [hipcc] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", sp, "-o", obj]
            if verbose:
                print("[longspec_amd.build]", " ".join(cmd), flush=True)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                raise subprocess.CalledProcessError(r.returncode, cmd)
            _write_usage(r.stderr, usage)
            # the compiler's own warnings reach the user; the resource-usage remarks (thousands of lines) go to the json
            noise = ("-Rpass-analysis=kernel-resource-usage", "remark:")
            for line in r.stderr.splitlines():
                if line.strip() and not any(n in line for n in noise) and "warning" in line:
                    sys.stderr.write(line + "\n")
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print("[longspec_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_variant(name: str, defines, force: bool = False, verbose: bool = True, sources=("attn.hip",)) -> str:
    """``_lib/liblongspec_hip_<name>.so``: the same library with extra ``-D`` flags on `sources` -- diagnostic and TEST builds
    only (loaded through LONGSPEC_HIP_LIB, never by default).  ``mutant`` (-DLS_MUTATE_SKIP_BLOCK) is the library
    tests/test_gpu_ops.py::test_mutant_is_caught expects the metric-size parity test to reject."""
    build(verbose=verbose)
    hipcc = _hipcc()
    out = os.path.join(LIBDIR, f"liblongspec_hip_{name}.so")
    objs = []
    for s_ in SOURCES:
        if s_ not in sources:
            objs.append(os.path.join(LIBDIR, s_.replace(".hip", ".o")))
            continue
        src = os.path.join(CSRC, s_)
        obj = os.path.join(LIBDIR, s_.replace(".hip", f"_{name}.o"))
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + FLAGS + list(defines) + ["-c", src, "-o", obj]
            if verbose:
                print("[longspec_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(out, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
