"""Attribute the small framework kernels of a tree round to the host lines that launch them
(torch profiler with stacks).  Run on the GPU box:  python tools/prof_round_ops.py [--rounds 6]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--prefix", type=int, default=16384)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = bench.make_config("llama3-8b-262k")
    max_gen = 6 * (args.rounds + 6) + 16
    m = bench.build_model(cfg, dev, 0.02, seed=1234)
    m.set_max_gen_len(max_gen + 256)
    m.glide.set_max_gen_len(max_gen + 256)
    bench.synth_kv(m, args.prefix, args.prefix, max_gen + 256, dev, seed=4321)
    lens = torch.tensor([args.prefix], dtype=torch.int32, device=dev)
    first = torch.tensor([1000], dtype=torch.int64, device=dev)
    from torch.profiler import ProfilerActivity, profile
    with torch.inference_mode():
        st = m.begin_tree_decode(first, lens, args.prefix, bench.TREE, max_gen, eos_id=-1)
        st.eos = None
        st.use_graphs = False          # profile the launch-by-launch path
        for _ in range(3):
            m.tree_round(st)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for _ in range(args.rounds):
                m.tree_round(st)
            torch.cuda.synchronize()
    # kernel events -> launching python frame inside longspec_amd
    by_site = collections.defaultdict(lambda: [0, 0.0, set()])
    evs = prof.events()
    for e in evs:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
            continue
        site = "?"
        for fr in (e.stack or []):
            if "longspec_amd" in fr and "_C.py" not in fr:
                site = fr.split("longspec_amd/")[-1]
                break
        for k in e.kernels:
            if any(s in k.name for s in ("skinny_gemm", "attn_partial", "attn_finish")):
                continue
            d = by_site[(site, e.name)]
            d[0] += 1
            d[1] += k.duration
            d[2].add(k.name[:60])
    rows = sorted(by_site.items(), key=lambda kv: -kv[1][1])
    tot = 0.0
    for (site, op), (n, us, names) in rows:
        tot += us
        print(f"{us / args.rounds:8.1f} us/round {n / args.rounds:5.1f}x  {op:28s} {site:48s} {sorted(names)[0]}")
    print(f"total {tot / args.rounds:.1f} us/round")


if __name__ == "__main__":
    main()
