"""A/B timing of a host-side switch on ONE box (box-to-box variation is ~5 %, larger than most remaining effects):
alternates blocks of tree rounds with the switch on and off.   python tools/ab_round.py llama.FUSE_QKV_ROPE"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    # "graphs": HIP-graph replay of the rounds on/off; otherwise <module>.<FLAG> of longspec_amd (rounds issued eagerly:
    # a captured graph would not see the flag change)
    graphs_ab = sys.argv[1] == "graphs"
    if not graphs_ab:
        modname, attr = sys.argv[1].rsplit(".", 1)
        mod = importlib.import_module("longspec_amd." + modname)
    else:
        attr = "graphs"
    blocks, per = 6, 15
    dev = torch.device("cuda", 0)
    cfg = bench.make_config("llama3-8b-262k")
    max_gen = 6 * (2 * blocks * per + 20) + 16
    m = bench.build_model(cfg, dev, 0.02, seed=1234)
    m.set_max_gen_len(max_gen + 256)
    m.glide.set_max_gen_len(max_gen + 256)
    bench.synth_kv(m, 16384, 16384, max_gen + 256, dev, seed=4321)
    lens = torch.tensor([16384], dtype=torch.int32, device=dev)
    first = torch.tensor([1000], dtype=torch.int64, device=dev)
    t = {True: [], False: []}
    with torch.inference_mode():
        st = m.begin_tree_decode(first, lens, 16384, bench.TREE, max_gen, eos_id=-1)
        st.eos = None

        def set_flag(flag):
            if graphs_ab:
                st.use_graphs = flag
            else:
                st.use_graphs = False
                setattr(mod, attr, flag)

        if graphs_ab:
            m.prepare_tree_graphs(st)
        for flag in (True, False):
            set_flag(flag)
            for _ in range(4):
                m.tree_round(st)
        for b in range(blocks):
            for flag in (True, False):
                set_flag(flag)
                m.tree_round(st)
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(per):
                    m.tree_round(st)
                torch.cuda.synchronize()
                t[flag].append((time.time() - t0) / per * 1e3)
    for flag in (True, False):
        v = sorted(t[flag])
        print(f"{attr}={flag}: median {v[len(v) // 2]:.4f} ms/round  min {v[0]:.4f}  all {[round(x, 3) for x in t[flag]]}")


if __name__ == "__main__":
    main()
