#!/usr/bin/env python3
"""Micro-benchmark of the hybrid verification attention kernel pair on one MI355X:
HIP-event timing on the launch stream, algorithmic bytes / time."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import toy
from longspec_amd import ops


def algo_bytes(L, H, Hkv, R=74, D=128):
    return 2 * L * Hkv * D * 2 + 2 * R * Hkv * D * 2 + 2 * R * H * D * 2 + R * R // 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, nargs="+", default=[16384, 131072])
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--Hkv", type=int, default=8)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--mode", default="verify", choices=["verify", "prefix"])
    ap.add_argument("--sq", type=int, default=74)
    args = ap.parse_args()
    dev = "cuda"
    for L in args.L:
        H, Hkv = args.H, args.Hkv
        q, k, v, _, _, tm = toy.verify_inputs(H, Hkv, 1, 1234)
        if args.sq != 74:
            q = toy.randn_f16((1, args.sq, H, 128), 7)
        gen = torch.Generator(device="cpu").manual_seed(1235)
        kc = torch.randn(1, L + 512, Hkv, 128, generator=gen).to(torch.float16).to(dev)
        vc = torch.randn(1, L + 512, Hkv, 128, generator=gen).to(torch.float16).to(dev)
        q, k, v = q.to(dev), k.to(dev), v.to(dev)
        bits = ops.pack_tree_mask(tm.to(dev))
        cl = torch.tensor([L], dtype=torch.int32, device=dev)

        def call():
            if args.mode == "verify":
                return ops.verify_attention(q, k, v, kc, vc, cl, bits, False, kv_len_hint=L, n_splits=args.splits)
            return ops.kvcache_attention(q, kc, vc, cache_seqlens=cl, kv_len_hint=L, n_splits=args.splits)

        for _ in range(5):
            call()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            call()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / args.iters
        by = algo_bytes(L, H, Hkv, R=args.sq)
        flops = 4 * args.sq * H * 128 * L
        print(json.dumps({"mode": args.mode, "L": L, "H": H, "Hkv": Hkv, "sq": args.sq, "us_per_call": round(us, 2),
                          "algo_GBps": round(by / us / 1e3, 1), "frac_of_8TBps": round(by / us / 1e3 / 8000, 4),
                          "TFLOPs": round(flops / us / 1e6, 1)}))


if __name__ == "__main__":
    main()
