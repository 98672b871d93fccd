"""One GEMM shape, ours and torch, a few calls each -- a target for rocprofv3 (kernel trace / PMC)."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from longspec_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=74)
ap.add_argument("--N", type=int, default=14336)
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--calls", type=int, default=6)
ap.add_argument("--splits", type=int, default=0)
ap.add_argument("--silu", action="store_true")
ap.add_argument("--no-torch", action="store_true")
ap.add_argument("--copies", type=int, default=4, help="1 = the same weight every call (served by the 256 MB Infinity Cache)")
a = ap.parse_args()
g = torch.Generator(device="cpu").manual_seed(0)
Ws = [(torch.randn(a.N, a.K, generator=g) * 0.02).half().cuda() for _ in range(4)]
NC = max(1, min(4, a.copies))
x = torch.randn(a.M, a.K, generator=g).half().cuda()
PW = [ops.pack_weight(w) for w in Ws]
PGU = [ops.pack_gate_up(Ws[i], Ws[(i + 1) % 4]) for i in range(4)] if a.silu else None
for i in range(a.calls):
    if a.silu:
        ops.mlp_gate_up(x, PGU[i % NC], n_splits=a.splits)
    else:
        ops.linear(x, PW[i % NC], n_splits=a.splits)
    if not a.no_torch:
        torch.nn.functional.linear(x, Ws[i % 4])
torch.cuda.synchronize()
