#!/usr/bin/env python3
"""Prompt attention (ops.prefill_attention: causal flash attention + cache fill through the decode kernels) timed alone:
python tools/bench_prefill.py [--L 32768 65536] [--H 32 --Hkv 8].  Reports tokens/s and the causal-attention TFLOP/s."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longspec_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, nargs="+", default=[16384, 65536])
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--Hkv", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda", 0)
for L in a.L:
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn((1, L, a.H, 128), generator=g, device=dev, dtype=torch.float16)
    k = torch.randn((1, L, a.Hkv, 128), generator=g, device=dev, dtype=torch.float16)
    v = torch.randn((1, L, a.Hkv, 128), generator=g, device=dev, dtype=torch.float16)
    kc = torch.zeros((1, L + 256, a.Hkv, 128), device=dev, dtype=torch.float16)
    vc = torch.zeros_like(kc)
    ops.prefill_attention(q[:, :2048], k[:, :2048], v[:, :2048], kc, vc)      # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    o = ops.prefill_attention(q, k, v, kc, vc)
    torch.cuda.synchronize()
    dt = time.time() - t0
    flops = 4.0 * L * L / 2 * a.H * 128
    print(json.dumps({"L": L, "H": a.H, "Hkv": a.Hkv, "seconds": round(dt, 4), "tokens_per_s": round(L / dt),
                      "TFLOPs_causal": round(flops / dt / 1e12, 1)}))
