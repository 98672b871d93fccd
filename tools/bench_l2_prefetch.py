"""What is an L2-resident head of the weight stream worth to a projection launch?  (round 4)

Before each timed ``ls_linear_fwd`` an ``ls_linear_prefetch`` launch of the same grid requests the first U register sets
(8 KB per wave each) of every wave's weights with the default cache policy; the projection itself is bracketed by events.
U = 0 is the plain launch.  Weights rotate over copies so that nothing is served from the 256 MB Infinity Cache.

    python tools/bench_l2_prefetch.py [--rows 74]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from longspec_amd import ops


def timed(fn, n=30):
    for _ in range(4):
        fn(None)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        b.record()                   # (creates the underlying events; the C ABI re-records them around the kernel)
    torch.cuda.synchronize()
    for a, b in evs:
        fn((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="*", default=[74, 1])
    args = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    dev = "cuda"
    nrot = 6
    for M in args.rows:
        x4 = torch.randn(M, 4096, generator=g).half().to(dev)
        x14 = torch.randn(M, 14336, generator=g).half().to(dev)
        cos = torch.ones(M, 128).half().to(dev)
        sin = torch.zeros(M, 128).half().to(dev)

        def mk(N, K, rope=False):
            return [ops.pack_weight((torch.randn(N, K, generator=g) * 0.02).half().to(dev), rope=rope) for _ in range(nrot)]
        Wo, Wd = mk(4096, 4096), mk(4096, 14336)
        Wq, Wk, Wv = mk(4096, 4096, True), mk(1024, 4096, True), mk(1024, 4096)
        Wg = [ops.pack_gate_up((torch.randn(14336, 4096, generator=g) * 0.02).half().to(dev),
                               (torch.randn(14336, 4096, generator=g) * 0.02).half().to(dev)) for _ in range(2)]
        i = [0]

        def nxt(ws):
            i[0] += 1
            return ws[i[0] % len(ws)]
        shapes = [("o_proj 33.5MB", lambda t: ops.linear(x4, nxt(Wo), timing=t)),
                  ("q|k|v+rope 50MB", lambda t: ops.linear_qkv_rope(x4, [nxt(Wq), Wk[i[0] % nrot], Wv[i[0] % nrot]], [None] * 3, cos, sin, timing=t)),
                  ("down 117MB", lambda t: ops.linear(x14, nxt(Wd), timing=t)),
                  ("gate|up 235MB", lambda t: ops.mlp_gate_up(x4, nxt(Wg), timing=t))]
        for name, fn in shapes:
            row = []
            for U in (0, 1, 3, 6, 12, 64):
                ops.PREFETCH_PROBE = U
                row.append((U, timed(fn)))
            ops.PREFETCH_PROBE = 0
            print(f"M={M:3d} {name:16s} " + "  ".join(f"U={u}: {t:6.1f}us" for u, t in row), flush=True)


if __name__ == "__main__":
    main()
