import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longspec_amd import ops
torch.manual_seed(0)
L = 16384
kc = torch.randn(1, L + 256, 8, 128, device="cuda", dtype=torch.float16)
vc = torch.randn(1, L + 256, 8, 128, device="cuda", dtype=torch.float16)
# rotate over several caches so that the MALL does not serve the stream
caches = [(torch.randn_like(kc), torch.randn_like(vc)) for _ in range(6)]
cl = torch.tensor([L], dtype=torch.int32, device="cuda")
def timeit(fn, n=60):
    for _ in range(6): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t)//2]*1e3
for sq in (1, 4, 16):
    q = torch.randn(1, sq, 32, 128, device="cuda", dtype=torch.float16)
    for S in (0, 8, 12, 16, 20, 24, 28, 31, 32, 40, 48, 64):
        i = [0]
        def f():
            i[0] = (i[0] + 1) % len(caches)
            k, v = caches[i[0]]
            return ops.kvcache_attention(q, k, v, cache_seqlens=cl, causal=False, kv_len_hint=L, n_splits=S)
        try:
            t = timeit(f)
            print(f"sq={sq:2d} n_splits={S:2d}: {t:6.1f} us  ({2*L*8*128*2/t/1e6:5.2f} TB/s)", flush=True)
        except Exception as e:
            print(f"sq={sq} S={S}: {type(e).__name__} {str(e)[:80]}")
