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
    ap.add_argument("--zeros", action="store_true", help="all-zero K/V/q: the same instruction stream with no operand toggling (how much of the time is the clock the power budget allows)")
    ap.add_argument("--gap-us", type=float, default=0.0, help="idle time between calls (a cooler chip clocks higher)")
    ap.add_argument("--round-like", type=int, default=0, metavar="MB",
                    help="between two attention calls stream MB megabytes through a copy kernel, the way a decode round puts ~190 us "
                         "of weight-streaming GEMMs between them (436 MB per layer for Llama-3-8B): the attention kernel then runs at "
                         "the clock it gets inside a round, not at the one a back-to-back loop of itself is throttled to")
    ap.add_argument("--kv-gap", type=int, default=None, metavar="BYTES",
                    help="carve K and V out of ONE buffer, V starting at (K's size rounded up to 2 MB) + BYTES: does the distance "
                         "between the two streams a workgroup reads in step matter to the HBM channels?")
    ap.add_argument("--head-major", action="store_true",
                    help="K/V stored [b, Hkv, S, 128] and passed as the permuted [b, S, Hkv, 128] view: every workgroup streams one "
                         "contiguous region instead of 256 B out of every 2 KB row")
    ap.add_argument("--score-scale", type=float, default=1.0,
                    help="multiply the prefix K by this factor: soft-max logits ~ N(0, scale^2) instead of N(0, 1) -- heavier tails, "
                         "more keys far above the maximum of a split's first 64 keys (the kernel's fixed soft-max reference)")
    ap.add_argument("--sink", action="store_true",
                    help="attention-sink + recency pattern: keys 0-3 score +12 nats for every query, and the logits ramp up by 6 nats "
                         "over the last 2048 keys of the prefix")
    ap.add_argument("--hot-keys", type=int, default=0, metavar="N",
                    help="N keys at pseudo-random positions score +14 nats for every query (a retrieval head): each lands behind "
                         "the first 64 keys of some split and leaves the fp16 range of that split's reference")
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
        if args.kv_gap is not None:
            nb = kc.numel() * 2
            v_at = (nb + (2 << 20) - 1) // (2 << 20) * (2 << 20) + args.kv_gap
            buf = torch.empty(v_at + nb, dtype=torch.uint8, device=dev)
            k2 = buf[:nb].view(torch.float16).view(kc.shape)
            v2 = buf[v_at:v_at + nb].view(torch.float16).view(vc.shape)
            k2.copy_(kc)
            v2.copy_(vc)
            kc, vc = k2, v2
        if args.head_major:
            kc = kc.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
            vc = vc.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
        q, k, v = q.to(dev), k.to(dev), v.to(dev)
        if args.score_scale != 1.0:
            kc[:, :L] *= args.score_scale
        if args.sink or args.hot_keys:
            # a direction every query shares: q <- q + a*u, k_hot <- k_hot + b*u with a*b*|u|^2*scale = the wanted logit offset
            u = torch.zeros(128, dtype=torch.float32)
            u[:16] = 1.0
            a = 2.0
            q = (q.float() + a * u.to(dev)).half()

            def bump(rows, nats):
                b = nats / (a * 16.0 / (128 ** 0.5))
                kc[0, rows] = (kc[0, rows].float() + b * u.to(dev)).half()
            if args.sink:
                bump(torch.arange(0, 4, device=dev), 12.0)
                ramp = torch.arange(max(0, L - 2048), L, device=dev)
                bq = (6.0 * (ramp - ramp[0]).float() / max(1, len(ramp) - 1)) / (a * 16.0 / (128 ** 0.5))
                kc[0, ramp] = (kc[0, ramp].float() + bq[:, None, None] * u.to(dev)).half()
            if args.hot_keys:
                gh = torch.Generator().manual_seed(99)
                pos = torch.randint(0, L, (args.hot_keys,), generator=gh).to(dev)
                bump(pos, 14.0)
        if args.zeros:
            for t in (q, k, v, kc, vc):
                t.zero_()
        bits = ops.pack_tree_mask(tm.to(dev))
        cl = torch.tensor([L], dtype=torch.int32, device=dev)

        def call():
            if args.mode == "verify":
                return ops.verify_attention(q, k, v, kc, vc, cl, bits, False, kv_len_hint=L, n_splits=args.splits)
            return ops.kvcache_attention(q, kc, vc, cache_seqlens=cl, kv_len_hint=L, n_splits=args.splits)

        for _ in range(5):
            call()
        torch.cuda.synchronize()
        from longspec_amd import _C
        _C.load().ls_attn_redo_count(1)
        call()
        redo_per_call = _C.load().ls_attn_redo_count(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if args.round_like > 0:
            src = torch.empty(args.round_like * (1 << 20) // 2, dtype=torch.uint8, device=dev)
            dst = torch.empty_like(src)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
            for _ in range(3):
                call()
                dst.copy_(src)
            for a_, b_ in evs:
                a_.record()
                call()
                b_.record()
                dst.copy_(src)
            torch.cuda.synchronize()
            us = sum(a_.elapsed_time(b_) for a_, b_ in evs) * 1e3 / args.iters
        elif args.gap_us > 0:
            import time
            tot = 0.0
            for _ in range(args.iters):
                s.record()
                call()
                e.record()
                torch.cuda.synchronize()
                tot += s.elapsed_time(e) * 1e3
                time.sleep(args.gap_us * 1e-6)
            us = tot / args.iters
        else:
            s.record()
            for _ in range(args.iters):
                call()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / args.iters
        by = algo_bytes(L, H, Hkv, R=args.sq)
        flops = 4 * args.sq * H * 128 * L
        print(json.dumps({"score_scale": args.score_scale, "sink": args.sink, "hot_keys": args.hot_keys,
                          "redo_workgroups_per_call": int(redo_per_call), "mode": args.mode, "kv_gap": args.kv_gap, "head_major": args.head_major, "zeros": args.zeros, "gap_us": args.gap_us, "round_like_MB": args.round_like, "L": L, "H": H, "Hkv": Hkv, "sq": args.sq, "us_per_call": round(us, 2),
                          "algo_GBps": round(by / us / 1e3, 1), "frac_of_8TBps": round(by / us / 1e3 / 8000, 4),
                          "TFLOPs": round(flops / us / 1e6, 1)}))


if __name__ == "__main__":
    main()
