#!/usr/bin/env python3
"""RMSNorm folded into the consuming projection against the chain it replaces, per layer of a verification pass
(Llama-3-8B dims, 74 rows): rmsnorm + q|k|v(+rope) vs q|k|v(norm=), rmsnorm + gate|up vs gate|up(norm=), o_proj / down_proj
with and without the sum-of-squares side output.  Weights rotate over COPIES distinct buffers so that no launch finds its
matrix in the Infinity Cache.   python tools/bench_norm_fold.py [--rows 74] [--copies 6] [--iters 30]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longspec_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=74)
    ap.add_argument("--copies", type=int, default=6)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--inter", type=int, default=14336)
    ap.add_argument("--kv", type=int, default=1024)
    a = ap.parse_args()
    dev, dt = "cuda", torch.float16
    M, Hd, I, C = a.rows, a.hidden, a.inter, a.copies
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(dt)
    qkv = [[ops.pack_weight(rnd(n, Hd, sc=Hd ** -0.5), rope=i < 2) for i, n in enumerate((Hd, a.kv, a.kv))] for _ in range(C)]
    gu = [ops.pack_gate_up(rnd(I, Hd, sc=Hd ** -0.5), rnd(I, Hd, sc=Hd ** -0.5)) for _ in range(C)]
    wo = [ops.pack_weight(rnd(Hd, Hd, sc=Hd ** -0.5)) for _ in range(C)]
    wd = [ops.pack_weight(rnd(Hd, I, sc=I ** -0.5)) for _ in range(C)]
    nw = (1 + 0.1 * torch.randn(Hd, device=dev, generator=g)).to(dt)
    h, res, act, attn = rnd(1, M, Hd), rnd(1, M, Hd), rnd(1, M, I), rnd(1, M, Hd)
    pos = torch.arange(1000, 1000 + M, device=dev)[None]
    inv_freq = (1.0 / (10000 ** (torch.arange(0, 128, 2).float() / 128))).to(dev)
    cos, sin = ops.rope_cos_sin(pos, inv_freq, 1.0, dt)
    _, ssq = ops.linear(attn, wo[0], None, residual=res, ssq_out=True)
    fold = ops.NormFold(nw, 1e-5, ssq)

    cases = {
        "rmsnorm + q|k|v(rope)": lambda i: ops.linear_qkv_rope(ops.rmsnorm(h, nw, 1e-5), qkv[i], [None] * 3, cos, sin),
        "q|k|v(rope, norm folded)": lambda i: ops.linear_qkv_rope(h, qkv[i], [None] * 3, cos, sin, norm=fold),
        "rmsnorm(+residual) + gate|up": lambda i: ops.mlp_gate_up(ops.rmsnorm(h, nw, 1e-5, residual=res)[0], gu[i]),
        "gate|up(norm folded)": lambda i: ops.mlp_gate_up(h, gu[i], norm=fold),
        "o_proj": lambda i: ops.linear(attn, wo[i], None),
        "o_proj(+residual, ssq_out)": lambda i: ops.linear(attn, wo[i], None, residual=res, ssq_out=True),
        "down_proj": lambda i: ops.linear(act, wd[i], None),
        "down_proj(+residual, ssq_out)": lambda i: ops.linear(act, wd[i], None, residual=res, ssq_out=True),
    }
    out = {}
    for name, fn in cases.items():
        for i in range(C):
            fn(i)
        torch.cuda.synchronize()
        best = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(a.iters):
                fn(it % C)
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3 / a.iters)
        out[name] = round(min(best), 2)
    print(json.dumps({"rows": M, "hidden": Hd, "inter": I, "us_per_call (eager launches, host-bound floor ~ 6 us per launch)": out}))
    # the same under one HIP graph per case (what the decode round replays)
    outg = {}
    for name, fn in cases.items():
        gph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn(0)
            torch.cuda.synchronize()
            with torch.cuda.graph(gph, stream=s):
                for it in range(C):
                    fn(it % C)
        torch.cuda.synchronize()
        best = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gph.replay()
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3 / (5 * C))
        outg[name] = round(min(best), 2)
    print(json.dumps({"us_per_call (replayed from a HIP graph)": outg}))


if __name__ == "__main__":
    main()
