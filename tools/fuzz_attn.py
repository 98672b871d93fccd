#!/usr/bin/env python3
"""Random-shape sweep of the attention entry points against the CPU oracle (oracle/ref_ops.py) -- a robustness tool, not part of
the test suite: head layouts, row counts, prefix lengths around the kernels' switch points (16-row tiles, 32-key blocks, the
4096-row threshold of the two-chunk path, split counts), dtypes, causal / window / append / verify modes.
    python tools/fuzz_attn.py [--cases 150] [--seed 0]"""
import argparse
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import toy  # noqa: E402
from longspec_amd import ops  # noqa: E402
from oracle import ref_ops  # noqa: E402

DEV = "cuda"


def close_abs(got, want, dtype, what):
    """The verification call's bar: one 16-bit ulp at the magnitude of the MERGE's operands (prefix_o * w and
    current_out * (1 - w) are rounded at magnitudes up to ~2, whatever their sum is): 1.1e-3 absolute in fp16."""
    d = (got.float().cpu() - want.float()).abs()
    atol, mean = (1.1e-3, 3.5e-4) if dtype == torch.float16 else (8.5e-3, 2.5e-3)
    # with a prefix of a few dozen keys the outputs themselves reach 2-4: the ulp of the merge's operands grows with them.
    # Per (row, head): operands up to twice the largest output of the row, rounded up to a power of two.
    top = (2.0 * want.float().abs().amax(-1, keepdim=True)).clamp_min(1.0)
    tol = atol * torch.exp2(torch.ceil(torch.log2(top)))
    # (a handful of elements per call may sit one more rounding step out: 3e5 elements, tie flips of 16-bit intermediates)
    ok = bool((d <= 2 * tol).all()) and int((d > tol).sum()) <= 4 and d.mean().item() <= mean and bool(torch.isfinite(got.float()).all())
    return ok, f"{what}: max |diff| {d.max().item():.2e} (worst {float((d / tol).max()):.2f} of its bound), mean {d.mean().item():.2e}"


def close(got, want, dtype, what):
    d = (got.float().cpu() - want.float()).abs()
    scale = want.float().abs().clamp_min(want.float().pow(2).mean().sqrt())
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    bad = d > (2.5 * ulp * scale + 1e-6)
    frac = bad.float().mean().item()
    worst = (d / (ulp * scale)).max().item()
    ok = worst <= 4.5 and frac <= 0.002 and bool(torch.isfinite(got.float()).all())
    return ok, f"{what}: worst {worst:.2f} ulp, {frac:.4%} beyond 2.5 ulp"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    fails = 0
    for case in range(a.cases):
        Hkv = rnd.choice([1, 2, 4, 8])
        grp = rnd.choice([1, 2, 4, 5, 8])
        H = Hkv * grp
        dtype = rnd.choice([torch.float16, torch.float16, torch.bfloat16])
        mode = rnd.choice(["prefix", "prefix", "causal", "window", "append", "verify", "verify"])
        L = rnd.choice([0, 1, 31, 32, 33, 63, 64, 65, 127, 300, 511, 512, 513, 1000, 2047, 4095, 4096, 4097, 4100, 6000, 9000])
        L = max(0, L + rnd.choice([0, 0, -1, 1, 7]))
        splits = rnd.choice([0, 0, 0, 1, 2, 3, 5, 8, 31, 32, 64])
        sq = 74 if mode == "verify" else rnd.choice([1, 2, 3, 4, 6, 16, 17, 37, 53, 74, 80])
        if mode in ("causal",) and L < sq:
            L = sq + L
        if mode == "window" and L < 1:
            L = 1
        if mode == "prefix" and L < 1:
            L = 1
        seed = 1000 + case
        q = toy.randn_f16((1, sq, H, 128), seed).to(dtype)
        kc = torch.zeros(1, L + 128, Hkv, 128, dtype=dtype)
        vc = torch.zeros_like(kc)
        kc[:, :L] = toy.randn_f16((1, L, Hkv, 128), seed + 1).to(dtype)
        vc[:, :L] = toy.randn_f16((1, L, Hkv, 128), seed + 2).to(dtype)
        cl = torch.tensor([L], dtype=torch.int32)
        desc = f"case {case}: {mode} H={H}/{Hkv} sq={sq} L={L} splits={splits} {str(dtype)[6:]}"
        try:
            if mode == "verify":
                qv, k, v, _, _, tm = toy.verify_inputs(H, Hkv, 1, seed, a=rnd.choice([1, 3, 6]))
                qv, k, v = qv.to(dtype), k.to(dtype), v.to(dtype)
                last = rnd.random() < 0.3
                kr, vr = kc.clone(), vc.clone()
                want = ref_ops.target_verify_attention(qv, k, v, kr, vr, cl, tm, last)
                kg, vg = kc.to(DEV), vc.to(DEV)
                got = ops.verify_attention(qv.to(DEV), k.to(DEV), v.to(DEV), kg, vg, cl.to(DEV), ops.pack_tree_mask(tm.to(DEV)), last,
                                           kv_len_hint=L, n_splits=splits)
                ok, msg = close_abs(got, want, dtype, "o")
                ok = ok and torch.equal(kg.cpu(), kr) and torch.equal(vg.cpu(), vr)
            elif mode == "append":
                n = min(sq, rnd.choice([1, 3, 6]))
                q = q[:, :n]
                k = toy.randn_f16((1, n, Hkv, 128), seed + 3).to(dtype)
                v = toy.randn_f16((1, n, Hkv, 128), seed + 4).to(dtype)
                win = rnd.choice([-1, 512])
                kr, vr = kc.clone(), vc.clone()
                want = ref_ops.kvcache_attention(q, kr, vr, k, v, cache_seqlens=cl, causal=True, window_size=(win, -1))
                kg, vg = kc.to(DEV), vc.to(DEV)
                got = ops.kvcache_attention(q.to(DEV), kg, vg, k.to(DEV), v.to(DEV), cache_seqlens=cl.to(DEV), causal=True,
                                            window_size=(win, -1), kv_len_hint=L, n_splits=splits)
                ok, msg = close(got, want, dtype, "o")
                ok = ok and torch.equal(kg.cpu(), kr) and torch.equal(vg.cpu(), vr)
            else:
                kw = {"causal": True} if mode == "causal" else {"window_size": (512, -1)} if mode == "window" else {}
                want, lse_w = ref_ops.kvcache_attention(q, kc, vc, cache_seqlens=cl, return_softmax_lse=True, **kw)
                got, lse = ops.kvcache_attention(q.to(DEV), kc.to(DEV), vc.to(DEV), cache_seqlens=cl.to(DEV), return_softmax_lse=True,
                                                 kv_len_hint=L + rnd.choice([0, 0, 100]), n_splits=splits, **kw)
                ok, msg = close(got, want, dtype, "o")
                dl = (lse.cpu() - lse_w)
                dl = dl[torch.isfinite(lse_w)].abs().max().item() if torch.isfinite(lse_w).any() else 0.0
                ok = ok and dl <= (3e-4 if dtype == torch.float16 else 2e-3)
                msg += f", lse {dl:.1e}"
        except Exception as e:      # noqa: BLE001
            ok, msg = False, f"{type(e).__name__}: {str(e)[:200]}"
        if not ok:
            fails += 1
            print("FAIL", desc, "--", msg, flush=True)
        elif case % 25 == 0:
            print("ok  ", desc, "--", msg, flush=True)
    print(f"{a.cases - fails}/{a.cases} cases passed")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
