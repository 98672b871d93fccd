"""The sequence-sharded decode path on the REAL kernels: two ranks share cuda:0 (one GPU box) and talk
over gloo; each holds half of the prefix KV of every target layer, every attention call is
ls_attn_partial -> ls_attn_reduce_local -> all-gather -> ls_attn_finish.  Token ids must equal the
single-process golden run."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, run_name, q, sharded_prefill=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from longspec_amd.dist import KVShard, shard_model_kv
    from longspec_amd.llama_glide import LlamaGlide
    run = [r for r in cases.generate_runs() if r["name"] == run_name][0]
    m = LlamaGlide(run["cfg"], device="cuda")
    m.load_state_dict({**run["target_sd"], **{"glide." + k: v for k, v in run["draft_sd"].items()}}, strict=True)
    P, glen = run["prompt_len"], run["max_gen_len"]
    ids = run["prompt"].cuda()
    if sharded_prefill:
        shard = KVShard(rank, world, shard_rows=(P + world - 1) // world)
        out, count, num, _, _ = m.tree_spec_generate(ids, torch.tensor([P], device="cuda"), tree_shape=run["tree_shape"],
                                                     max_gen_len=glen, eos_id=run["eos_id"], shard=shard)
        torch.cuda.synchronize()
        q.put((rank, out.cpu(), int(count), int(num)))
        dist.barrier()
        dist.destroy_process_group()
        return
    with torch.inference_mode():
        m.set_max_gen_len(glen + 256)
        m.glide.set_max_gen_len(glen + 256)
        m._set_hints(P, P)
        h = m.model.forward(ids, exec_type="prefill").last_hidden_state
        first = m.lm_head(h[:, P - 1]).argmax(-1)
        lens = torch.tensor([P], dtype=torch.int32, device="cuda")
        emb = m.model.embed_tokens(ids)
        pe = m.model.rotary_emb(emb, torch.arange(P, device="cuda")[None])
        m.glide(hidden_states=emb, position_embeddings=pe, llm_kv=m._last_kv(), cache_lens=lens.clone(),
                llm_kv_len=lens.clone(), exec_type="prefill")
        shard = KVShard(rank, world, shard_rows=(P + world - 1) // world)
        shard_model_kv(m, shard, P)
        st = m.begin_tree_decode(first, lens, P, run["tree_shape"], glen, 151645)
        for _ in range(1, glen):
            if not m.tree_round(st):
                break
    torch.cuda.synchronize()
    q.put((rank, st.output_ids.cpu(), int(st.count), int(st.num)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("run_name", ["mixed", "gqa_mixed"])
def test_sharded_tree_decode_on_gpu_matches_golden(run_name):
    world = 2
    run = [r for r in cases.generate_runs() if r["name"] == run_name][0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, run_name, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    for rank, out, count, num in res:
        assert torch.equal(out, run["tree_out"]), f"rank {rank}: token ids differ from the single-process reference"
        assert (count, num) == (run["tree_count"], run["tree_num"])


@pytest.mark.parametrize("run_name", ["mixed", "gqa_mixed"])
def test_sharded_prefill_and_decode_on_gpu_match_golden(run_name):
    """``tree_spec_generate(..., shard=...)``: sequence-sharded prefill (K/V all-gather per layer, rank-local causal
    blocks through the decode kernels) + sharded decode, two ranks on one GPU."""
    world = 2
    run = [r for r in cases.generate_runs() if r["name"] == run_name][0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, run_name, q, True)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    for rank, out, count, num in res:
        assert torch.equal(out, run["tree_out"]), f"rank {rank}: token ids differ from the single-process reference"
        assert (count, num) == (run["tree_count"], run["tree_num"])
