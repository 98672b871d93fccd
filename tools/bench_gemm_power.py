#!/usr/bin/env python3
"""Is the weight-streaming GEMM limited by the clock the power budget grants (as the attention kernel is: profiles/r3_power_bound.json)?
The same launches on random and on all-zero weights / activations, back to back, rotating over weight copies beyond the MALL."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from longspec_amd import ops


def run(name, N, K, M, zeros, gate_up=False, n=60):
    g = torch.Generator(device="cpu").manual_seed(0)
    copies = 4 if N * K * 2 < 300e6 else 2
    mk = (lambda *s: torch.zeros(*s).half().cuda()) if zeros else (lambda *s: (torch.randn(*s, generator=g) * 0.02).half().cuda())
    if gate_up:
        PW = [ops.pack_gate_up(mk(N, K), mk(N, K)) for _ in range(copies)]
        call = lambda w: ops.mlp_gate_up(x, w)
        nbytes = 2 * N * K * 2
    else:
        PW = [ops.pack_weight(mk(N, K)) for _ in range(copies)]
        call = lambda w: ops.linear(x, w)
        nbytes = N * K * 2
    x = torch.zeros(M, K).half().cuda() if zeros else torch.randn(M, K, generator=g).half().cuda()
    for i in range(8):
        call(PW[i % copies])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        call(PW[i % copies])
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / n
    print(json.dumps({"launch": name, "M": M, "zeros": zeros, "us": round(us, 2), "TBps": round(nbytes / us / 1e6, 3)}), flush=True)


for zeros in (False, True, False, True):
    run("gate|up+SiLU 2x14336x4096", 14336, 4096, 74, zeros, gate_up=True)
    run("down_proj 4096x14336", 4096, 14336, 74, zeros)
    run("lm_head 128256x4096", 128256, 4096, 74, zeros)
    run("lm_head 128256x4096", 128256, 4096, 16, zeros)
    run("o_proj 4096x4096", 4096, 4096, 74, zeros)
