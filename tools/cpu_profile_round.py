"""Host-side cost of a tree round: cProfile over N rounds (GPU work is asynchronous, so cumulative times are
CPU time spent issuing the round).   python tools/cpu_profile_round.py [--shard-path]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    shard = "--shard-path" in sys.argv
    dev = torch.device("cuda", 0)
    cfg = bench.make_config("llama3-8b-262k")
    rounds = 20
    max_gen = 6 * (rounds + 20) + 16
    m = bench.build_model(cfg, dev, 0.02, seed=1234)
    m.set_max_gen_len(max_gen + 256)
    m.glide.set_max_gen_len(max_gen + 256)
    bench.synth_kv(m, 16384, 16384, max_gen + 256, dev, seed=4321)
    if shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from longspec_amd.dist import KVShard
        sh = KVShard(0, 1, shard_rows=16384)
        for layer in m.model.layers:
            layer.self_attn.shard = sh
        m.glide.cross_attn.shard = sh
    lens = torch.tensor([16384], dtype=torch.int32, device=dev)
    first = torch.tensor([1000], dtype=torch.int64, device=dev)
    with torch.inference_mode():
        st = m.begin_tree_decode(first, lens, 16384, bench.TREE, max_gen, eos_id=-1)
        st.eos = None
        st.use_graphs = False          # profile the launch-by-launch path
        for _ in range(5):
            m.tree_round(st)
        torch.cuda.synchronize()
        # pure issue time: how long the host needs per round when it never waits for the GPU... the round's one host
        # read (state.tolist()) does wait, so also report the wall time
        pr = cProfile.Profile()
        t0 = time.time()
        pr.enable()
        for _ in range(rounds):
            m.tree_round(st)
        pr.disable()
        torch.cuda.synchronize()
        wall = (time.time() - t0) / rounds * 1e3
    print(f"wall {wall:.3f} ms/round")
    ps = pstats.Stats(pr)
    ps.sort_stats("tottime")
    ps.print_stats(28)


if __name__ == "__main__":
    main()
