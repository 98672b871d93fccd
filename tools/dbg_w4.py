import os, sys, subprocess, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from longspec_amd import ops
H, Hkv = int(os.environ.get("H", 4)), int(os.environ.get("HKV", 1))
res = {}
for L, S in [(32, 1), (64, 1), (96, 1), (128, 1), (300, 0), (300, 1), (1024, 1), (1024, 4), (4096, 0), (16384, 0)]:
    g = torch.Generator().manual_seed(L)
    q = torch.randn(1, 74, H, 128, generator=g).half().cuda()
    kc = torch.randn(1, L + 64, Hkv, 128, generator=g).half().cuda()
    vc = torch.randn(1, L + 64, Hkv, 128, generator=g).half().cuda()
    cl = torch.tensor([L], dtype=torch.int32, device="cuda")
    o, lse = ops.kvcache_attention(q, kc, vc, cache_seqlens=cl, return_softmax_lse=True, kv_len_hint=L, n_splits=S)
    torch.cuda.synchronize()
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), kc[:, :L].float().repeat_interleave(H // Hkv, 2)) / 128 ** 0.5
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vc[:, :L].float().repeat_interleave(H // Hkv, 2))
    res[f"L={L},S={S}"] = (round((o.float() - ref).abs().max().item(), 5), round((lse - torch.logsumexp(s, -1)).abs().max().item(), 6))
print(os.environ.get("LS_ATTN_KERNEL"), json.dumps(res))
