#!/usr/bin/env python3
"""Where a trip of attn_verify_kernel goes: s_memtime stamps of one wave of workgroup (split 1, kv head 0), taken
by a diagnostic build of the library (-DLS_V2_STAMPS), prefix-only call at L = 131072.

    python tools/v2_stamps.py build      # here: builds longspec_amd/_lib/liblongspec_hip_stamps<W>.so (W = stamped wave)
    LONGSPEC_HIP_LIB=... python tools/v2_stamps.py run   # on the GPU box
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = ["wait(vmcnt,lgkm)", "barrier", "dma issue", "phase A1", "phase B1", "phase A2", "phase B2", "loop back"]


def build():
    from longspec_amd import build as b
    exps = [int(x) for x in os.environ.get("V2_EXPS", "0").split(",")]
    for w in exps:
        objs = []
        for src in b.SOURCES:
            obj = os.path.join(b.LIBDIR, src.replace(".hip", ".o"))
            if src == "attn.hip":
                obj = os.path.join(b.LIBDIR, f"attn_stamps{w}.o")
                subprocess.check_call([b._hipcc()] + b.FLAGS + ["-DLS_V2_STAMPS", f"-DV2_EXP={w}", "-c", os.path.join(b.CSRC, src), "-o", obj])
            objs.append(obj)
        out = os.path.join(b.LIBDIR, f"liblongspec_hip_stamps{w}.so")
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
        print(out)


def run():
    import torch
    from longspec_amd import ops
    L, H, Hkv = int(os.environ.get("L", "131072")), 32, 8
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(1)
    q = torch.randn(1, 74, H, 128, generator=g).half().to(dev)
    kc = torch.randn(1, L + 64, Hkv, 128, generator=g).half().to(dev)
    vc = torch.randn(1, L + 64, Hkv, 128, generator=g).half().to(dev)
    cl = torch.tensor([L], dtype=torch.int32, device=dev)
    for _ in range(3):
        ops.kvcache_attention(q, kc, vc, cache_seqlens=cl, kv_len_hint=L)
    torch.cuda.synchronize()
    ws = max(ops.workspace_tensors(), key=lambda t: t.numel())
    # the stamps sit at the start of the new_o region: parts_o, parts_lse come first (256-byte aligned takes)
    n_parts = 32
    rows = 74 * H
    off = ((n_parts * rows * 128 * 4 + 255) // 256 * 256) + ((n_parts * rows * 4 + 255) // 256 * 256)
    raw = ws.view(torch.uint8)[off:off + 256 * 8].cpu().view(torch.int64).reshape(32, 8)
    rows_ = []
    for t in range(32):
        st = raw[t].tolist()
        nxt = raw[t + 1][0].item() if t + 1 < 32 else None
        d = [st[i + 1] - st[i] for i in range(7)] + [(nxt - st[7]) if nxt else 0]
        rows_.append(d)
    import statistics
    med = [statistics.median(r[i] for r in rows_[:-1]) for i in range(8)]
    print(json.dumps({"L": L, "segments": NAMES, "median_ticks": med, "trip_ticks": sum(med),
                      "note": "s_memtime ticks (100 MHz constant clock if not shader clock: compare ratios)",
                      "first_trips": rows_[:4]}))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
