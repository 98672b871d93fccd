#!/usr/bin/env python3
"""Wall-clock segment profile of attn_partial_pp_kernel (diagnostic build -DLS_PP_PROF): all 8 waves of workgroup (split 1,
kv head 0), prefix-only call at L (default 131072).  s_memrealtime ticks are 10 ns.
    python tools/build_variant.py ppprof -DLS_PP_PROF ; LONGSPEC_HIP_LIB=.../liblongspec_hip_ppprof.so python tools/pp_prof.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from longspec_amd import ops
L, H, Hkv = int(os.environ.get("L", "131072")), 32, 8
g = torch.Generator(device="cpu").manual_seed(1)
q = torch.randn(1, 74, H, 128, generator=g).half().cuda()
kc = torch.randn(1, L + 64, Hkv, 128, generator=g).half().cuda()
vc = torch.randn(1, L + 64, Hkv, 128, generator=g).half().cuda()
cl = torch.tensor([L], dtype=torch.int32, device="cuda")
for _ in range(3):
    ops.kvcache_attention(q, kc, vc, cache_seqlens=cl, kv_len_hint=L)
torch.cuda.synchronize()
ws = max(ops.workspace_tensors(), key=lambda t: t.numel())
n_parts = 32
rows = 74 * H
off = ((n_parts * rows * 128 * 4 + 255) // 256 * 256) + ((n_parts * rows * 4 + 255) // 256 * 256)
raw = ws.view(torch.uint8)[off:off + 64 * 8].cpu().view(torch.int64).tolist()
names = ["V body (DMA issue, V^T reads, soft-max)", "vmcnt wait (waves 0-3)", "barrier behind V", "M body (K reads, P.V, QK^T)",
         "vmcnt wait (waves 4-7)", "barrier behind M"]
out = {}
for w in range(8):
    r = raw[w * 8:w * 8 + 8]
    nb = max(r[6], 1)
    out[f"wave {w} (QT {r[7]})"] = {"steps": r[6], "ns_per_step": {n: round(r[i] * 10.0 / nb, 1) for i, n in enumerate(names)},
                                    "total_ns_per_step": round(sum(r[:6]) * 10.0 / nb, 1)}
print(json.dumps({"L": L, **out}, indent=1))
