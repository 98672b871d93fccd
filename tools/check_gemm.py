"""Correctness sweep of ls_linear_fwd over split counts and row counts (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from longspec_amd import ops

g = torch.Generator(device="cpu").manual_seed(1)
for (N, K) in [(4096, 4096), (14336, 4096), (1024, 4096), (512, 896), (4096, 14336), (100, 192)]:
    W = (torch.randn(N, K, generator=g) * 0.02).half().cuda()
    PW = ops.pack_weight(W)
    for M in (1, 5, 16, 17, 32, 74, 80):
        x = torch.randn(M, K, generator=g).half().cuda()
        ref = x.double() @ W.double().t()
        for S in (1, 2, 3, 5, 8):
            y = ops.linear(x, PW, n_splits=S)
            torch.cuda.synchronize()
            err = (y.double() - ref).abs().max().item()
            bad = (~torch.isfinite(y)).sum().item()
            flag = "" if err < 8e-3 and bad == 0 else "   <<<<<< BAD"
            if flag or M in (1, 74):
                print(f"N={N} K={K} M={M} S={S}: maxerr {err:.3e} nonfinite {bad}{flag}", flush=True)
for (N, K) in [(14336, 4096), (512, 256), (1024, 896)]:
    Wg = (torch.randn(N, K, generator=g) * 0.02).half().cuda()
    Wu = (torch.randn(N, K, generator=g) * 0.02).half().cuda()
    PGU = ops.pack_gate_up(Wg, Wu)
    for M in (1, 16, 30, 74):
        x = torch.randn(M, K, generator=g).half().cuda()
        want = torch.nn.functional.silu((x.double() @ Wg.double().t()).half().float()).half().float() * (x.double() @ Wu.double().t()).half().float()
        for S in (0, 1, 3):
            got = ops.mlp_gate_up(x, PGU, n_splits=S).float()
            err = (got - want).abs().max().item()
            print(f"silu N={N} K={K} M={M} S={S}: maxerr {err:.3e}" + ("   <<<<<< BAD" if not err < 1.6e-2 else ""), flush=True)
Wq = [(torch.randn(n, 4096, generator=g) * 0.02).half().cuda() for n in (4096, 1024, 1024)]
bq = [(torch.randn(n, generator=g) * 0.02).half().cuda() for n in (4096, 1024, 1024)]
PWq = [ops.pack_weight(w) for w in Wq]
for M in (1, 16, 74):
    x = torch.randn(M, 4096, generator=g).half().cuda()
    outs = ops.linear_multi(x, PWq, bq)
    for o, w, b in zip(outs, Wq, bq):
        ref = x.double() @ w.double().t() + b.double()
        err = (o.double() - ref).abs().max().item()
        print(f"multi M={M} n={w.shape[0]}: maxerr {err:.3e}" + ("   <<<<<< BAD" if not err < 8e-3 else ""), flush=True)
print("done")
