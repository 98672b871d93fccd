"""Kernel mix of the chain-speculation loop (`spec_generate`, gamma = 4): which launches are not the model's own
kernels.   python tools/prof_chain_ops.py [--prompt 4096] [--gen 96]"""
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
    ap.add_argument("--prompt", type=int, default=4096)
    ap.add_argument("--gen", type=int, default=96)
    ap.add_argument("--method", default="chain", choices=["chain", "vanilla", "magicdec"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = bench.make_config("llama3-8b-262k")
    m = bench.build_model(cfg, dev, 0.02, seed=1234)
    m.GRAPH_ROUNDS = False
    g = torch.Generator(device=dev).manual_seed(99)
    ids = torch.randint(5, cfg.vocab_size - 5, (1, args.prompt), generator=g, device=dev)
    pl = torch.tensor([args.prompt], device=dev)
    fn = {"chain": lambda: m.spec_generate(ids, pl, gamma=4, max_gen_len=args.gen, eos_id=-1),
          "vanilla": lambda: m.vanilla_generate(ids, pl, max_gen_len=args.gen, eos_id=-1),
          "magicdec": lambda: m.magicdec_generate(ids, pl, gamma=4, max_gen_len=args.gen)}[args.method]
    fn()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        r = fn()
        torch.cuda.synchronize()
    rounds = int(r[2]) if args.method != "vanilla" else int(r[1])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            a = agg[e.name[:90]]
            a[0] += 1
            a[1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
    tot = sum(v[1] for v in agg.values())
    print(f"{args.method}: {rounds} rounds/steps, decode {r[-2] if args.method != 'vanilla' else r[2]:.3f} s (incl. prefill kernels below)")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{us / rounds:9.1f} us/round {n / rounds:7.1f}x {us / n:8.2f} us  {k}")
    print(f"total {tot / rounds:.1f} us/round")


if __name__ == "__main__":
    main()
