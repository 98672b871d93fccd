#!/usr/bin/env python3
"""Wall-clock phase profile of attn_partial_ws_kernel (diagnostic build -DLS_WS_PROF): one S wave and one O wave of
workgroup (split 1, kv head 0), prefix-only call at L (default 131072).  s_memrealtime ticks are 10 ns.
    python tools/build_variant.py wsprof -DLS_WS_PROF ; LONGSPEC_HIP_LIB=.../liblongspec_hip_wsprof.so python tools/ws_prof.py"""
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
from longspec_amd import _C
import ctypes
_d = ops._desc(q, kc, vc, cl, L)
n_parts = _C.load().ls_attn_num_parts(ctypes.byref(_d))       # (31 since round 3: an odd split stride for calls without a new-key block)
rows = 74 * H
off = ((n_parts * rows * 128 * 4 + 255) // 256 * 256) + ((n_parts * rows * 4 + 255) // 256 * 256)
raw = ws.view(torch.uint8)[off:off + 32 * 8].cpu().view(torch.int64).tolist()
names = ["dma issue", "body (LDS reads, MFMA, soft-max / P.V)", "vmcnt wait (ladder + landing)", "lgkmcnt(0)", "barrier"]
out = {}
for role, base in (("S wave", 0), ("O wave", 8)):
    nb = raw[base + 5]
    out[role] = {"steps": nb, "ns_per_step": {n: round(raw[base + i] * 10.0 / max(nb, 1), 1) for i, n in enumerate(names)},
                 "total_ns_per_step": round(sum(raw[base:base + 5]) * 10.0 / max(nb, 1), 1)}
ms, mo = raw[16:24], raw[24:32]
tl = {"S wave (us from its entry)": {"K/V blocks 0-1 landed": (ms[1] - ms[0]) / 100, "reference look done": (ms[2] - ms[0]) / 100,
                                     "step loop done": (ms[3] - ms[0]) / 100, "after the redo check": (ms[4] - ms[0]) / 100},
      "O wave (us from its entry)": {"step loop done": (mo[3] - mo[0]) / 100, "ready to write the partial": (mo[5] - mo[0]) / 100,
                                     "partial stores acknowledged": (mo[6] - mo[0]) / 100}}
print(json.dumps({"L": L, **out, "timeline": tl}))
