#!/usr/bin/env python3
"""the warp-specialised verification kernel sits at 0.6-0.98 of the full-size parity bound
(tests/test_gpu_ops.py::assert_close_rel: |diff| <= 2^-11 (2 |ref| + 5 rms(ref))) on the ONE seed the tests use.  This sweeps
seeds through the same comparison (HIP path vs the C oracle, element by element) and records the distribution of the worst
ratio, so that "one seed from red" is a number:  python tools/seed_sweep_margins.py [--seeds 8] > gpurun_out/seed_sweep.json"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import toy                                                        # noqa: E402
from longspec_amd import ops                                      # noqa: E402
from oracle import c_port                                         # noqa: E402


def one(H, Hkv, L, last_layer, seed):
    q, k, v, _, _, tm = toy.verify_inputs(H, Hkv, 1, 4000 + L % 97 + 1000 * seed, a=4)
    gen = torch.Generator(device="cpu").manual_seed(L + 7919 * seed)
    kc = torch.zeros(1, L + 128, Hkv, 128, dtype=torch.float16)
    vc = torch.zeros(1, L + 128, Hkv, 128, dtype=torch.float16)
    kc[:, :L] = torch.randn(1, L, Hkv, 128, generator=gen).half()
    vc[:, :L] = torch.randn(1, L, Hkv, 128, generator=gen).half()
    ref = c_port.verify_attention(q, k, v, kc.clone(), vc.clone(), L, tm, last_layer).float()
    cl = torch.tensor([L], dtype=torch.int32)
    out = ops.verify_attention(q.cuda(), k.cuda(), v.cuda(), kc.cuda(), vc.cuda(), cl.cuda(), ops.pack_tree_mask(tm.cuda()),
                               last_layer, kv_len_hint=L).float().cpu()
    rms = ref.pow(2).mean().sqrt().item()
    d = (out - ref).abs()
    tol = 2.0 ** -11 * (2.0 * ref.abs() + 5.0 * rms)
    return dict(seed=seed, worst_ratio=round((d / tol).max().item(), 4), max_abs=d.max().item(),
                mean_over_ulp_rms=round(d.mean().item() / (2.0 ** -11 * rms), 4), rms=rms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    a = ap.parse_args()
    res = {}
    for (H, Hkv, L, last) in ((32, 8, 16384 + 37, True), (32, 32, 4096, False), (32, 8, 131072, False)):
        rows = [one(H, Hkv, L, last, s) for s in range(a.seeds)]
        res[f"H={H}/{Hkv} L={L} last={last}"] = dict(
            runs=rows, worst=max(r["worst_ratio"] for r in rows), best=min(r["worst_ratio"] for r in rows),
            max_abs=max(r["max_abs"] for r in rows))
    res["bound"] = "|diff| <= 2^-11 (2 |ref| + 5 rms(ref)) per element (tests/test_gpu_ops.py::assert_close_rel); north_star asks 1e-3 absolute"
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
