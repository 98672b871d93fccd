"""Draft cross-attention / vanilla decode attention over a long prefix: us and TB/s per query shape and split count.
    python tools/sweep_cross_attn_128k.py [L]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longspec_amd import ops

torch.manual_seed(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
caches = [(torch.randn(1, L + 256, 8, 128, device="cuda", dtype=torch.float16),
           torch.randn(1, L + 256, 8, 128, device="cuda", dtype=torch.float16)) for _ in range(3)]
cl = torch.tensor([L], dtype=torch.int32, device="cuda")


def timeit(fn, n=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3


for sq in (1, 3, 4, 16):
    q = torch.randn(1, sq, 32, 128, device="cuda", dtype=torch.float16)
    for S in [int(x) for x in os.environ.get("SWEEP_SPLITS", "0,16,24,31,32,48,64,96,128").split(",")]:
        i = [0]

        def f():
            i[0] = (i[0] + 1) % len(caches)
            k, v = caches[i[0]]
            return ops.kvcache_attention(q, k, v, cache_seqlens=cl, causal=False, kv_len_hint=L, n_splits=S)
        try:
            t = timeit(f)
            print(f"L={L} sq={sq:2d} n_splits={S:3d}: {t:6.1f} us  ({2 * L * 8 * 128 * 2 / t / 1e6:5.2f} TB/s)", flush=True)
        except Exception as e:
            print(f"sq={sq} S={S}: {type(e).__name__} {str(e)[:80]}")
