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


def _find_run(run_name):
    fam = "llama_long" if run_name.startswith("long_") else "llama"
    return [r for r in cases.generate_runs(fam) if r["name"] == run_name][0]


def _worker(rank, world, port, run_name, q, sharded_prefill=False, peer=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from longspec_amd.dist import KVShard, shard_model_kv
    from longspec_amd.llama_glide import LlamaGlide
    run = _find_run(run_name)
    m = LlamaGlide(run["cfg"], device="cuda")
    m.load_state_dict({**run["target_sd"], **{"glide." + k: v for k, v in run["draft_sd"].items()}}, strict=True)
    P, glen = run["prompt_len"], run["max_gen_len"]
    ids = run["prompt"].cuda()
    if sharded_prefill:
        shard = KVShard(rank, world, shard_rows=(P + world - 1) // world)
        out, count, num, _, _ = m.tree_spec_generate(ids, torch.tensor([P], device="cuda"), tree_shape=run["tree_shape"],
                                                     max_gen_len=glen, eos_id=run["eos_id"], shard=shard)
        torch.cuda.synchronize()
        timed_out = shard.peer.status()[1] if shard.peer is not None else None
        q.put((rank, out.cpu(), int(count), int(num), shard.peer is not None, timed_out))
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
        if peer:
            assert shard.enable_peer_exchange(128 * run["cfg"].num_attention_heads * 129, torch.device("cuda", 0))
        st = m.begin_tree_decode(first, lens, P, run["tree_shape"], glen, 151645)
        assert st.use_graphs == peer, "graphs replay under a shard exactly when its exchange is the peer-store one"
        for _ in range(1, glen):
            if not m.tree_round(st):
                break
    torch.cuda.synchronize()
    timed_out = shard.peer.status()[1] if shard.peer is not None else None
    q.put((rank, st.output_ids.cpu(), int(st.count), int(st.num), shard.peer is not None, timed_out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("peer", [False, True], ids=["all_gather", "peer_store_graphs"])
@pytest.mark.parametrize("run_name", ["mixed", "gqa_mixed"])
def test_sharded_tree_decode_on_gpu_matches_golden(run_name, peer):
    """peer=True: the exchange is csrc/xgmi.hip (IPC mailboxes of the two processes on the one GPU) and the rounds
    replay from HIP graphs; peer=False: the torch.distributed all-gather, launch by launch."""
    world = 2
    run = [r for r in cases.generate_runs() if r["name"] == run_name][0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, run_name, q, False, peer)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    for rank, out, count, num, has_peer, timed_out in res:
        assert has_peer == peer and not timed_out
        assert torch.equal(out, run["tree_out"]), f"rank {rank}: token ids differ from the single-process reference"
        assert (count, num) == (run["tree_count"], run["tree_num"])


# long_*: the regime of the BASELINE configurations on the sharded path -- prompts of 777 / 801 tokens split over
# two ranks (each rank's slice is shorter than the draft's 512-row window: the window spans the shard boundary), 96 / 89 rounds
# replayed from HIP graphs with the peer-store exchange.  These two runs reproduce the reference's count / num exactly on one GPU.
@pytest.mark.parametrize("run_name", ["mixed", "gqa_mixed", "long_mixed_s1", "long_gqa_s1"])
def test_sharded_prefill_and_decode_on_gpu_match_golden(run_name):
    """``tree_spec_generate(..., shard=...)``: sequence-sharded prefill (K/V all-gather per layer, rank-local causal
    blocks through the decode kernels) + sharded decode, two ranks on one GPU."""
    world = 2
    run = _find_run(run_name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, run_name, q, True)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    for rank, out, count, num, has_peer, timed_out in res:
        assert has_peer and not timed_out, "tree_spec_generate(shard=...) on a GPU maps the peer-store exchange by itself"
        assert torch.equal(out, run["tree_out"]), f"rank {rank}: token ids differ from the single-process reference"
        assert (count, num) == (run["tree_count"], run["tree_num"])


def _xchg_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from longspec_amd.dist import PeerExchange
    cap = 74 * 32 * 129
    px = PeerExchange(rank, world, cap, dev)
    ok = px.self_check()

    def record(r, i, n):
        return torch.arange(n, dtype=torch.float32, device=dev) * (r + 1) + 1000.0 * i

    bad = 0
    sizes = [4, 64, 4096, 4100, cap // 4 * 4, 128 * 129 * 4]
    for i in range(60):                              # eager, records of changing size (target pass / draft passes)
        n = sizes[i % len(sizes)]
        got = torch.zeros((world, n + 8), dtype=torch.float32, device=dev)       # a padded stride
        px.all_gather(record(rank, i, n), got)
        for r in range(world):
            bad += int(not torch.equal(got[r, :n], record(r, i, n)))
        bad += int(float(got[:, n:].abs().sum()) != 0.0)
    # three exchanges captured once and replayed: the epoch counter lives on the device
    n = 4096
    send = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(3)]
    recv = [torch.zeros((world, n), dtype=torch.float32, device=dev) for _ in range(3)]
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for k in range(3):
            px.all_gather(send[k], recv[k])          # warm-up, as every rank does
    torch.cuda.current_stream().wait_stream(st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(3):
            px.all_gather(send[k], recv[k])
    for rep in range(5):
        for k in range(3):
            send[k].copy_(record(rank, 100 + rep * 3 + k, n))
        g.replay()
        torch.cuda.synchronize()
        for k in range(3):
            for r in range(world):
                bad += int(not torch.equal(recv[k][r], record(r, 100 + rep * 3 + k, n)))
    done, timed_out = px.status()
    q.put((rank, ok, bad, done, timed_out))
    dist.barrier()
    px.close()
    dist.destroy_process_group()


def test_peer_exchange_two_processes_one_gpu():
    """ls_xchg_*: mailboxes mapped through hipIpc between two processes, eager and from a replayed HIP graph; every record
    arrives intact in rank order, nothing is written beyond n_floats, no wait times out."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xchg_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    for rank, ok, bad, done, timed_out in res:
        assert ok and bad == 0 and not timed_out, (rank, ok, bad, timed_out)
        assert done == 3 + 60 + 3 + 15, "self-check + eager + warm-up + 5 replays of 3"


def _silent_peer_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import time
    from longspec_amd import _C
    from longspec_amd.dist import PeerExchange
    px = PeerExchange(rank, world, 4096, dev)
    ok = px.self_check()
    _C.check(px.lib.ls_xchg_set_timeout(px._x, 0.2), "ls_xchg_set_timeout")
    took, timed_out, second = 0.0, False, 0.0
    if rank == 0:                                    # rank 1 never joins this exchange
        send = torch.ones(4096, dtype=torch.float32, device=dev)
        recv = torch.zeros((world, 4096), dtype=torch.float32, device=dev)
        t0 = time.time()
        px.all_gather(send, recv)
        torch.cuda.synchronize()
        took = time.time() - t0
        timed_out = px.status()[1]
        t0 = time.time()
        px.all_gather(send, recv)                    # latched: a second wait returns at once
        torch.cuda.synchronize()
        second = time.time() - t0
    q.put((rank, ok, took, timed_out, second))
    dist.barrier()                                   # rank 1 keeps its mailbox mapped until rank 0 is through
    px.close()
    dist.destroy_process_group()


def test_a_wait_on_a_silent_peer_gives_up_and_latches():
    """Failure detection of the peer exchange: a rank whose peer never pushes does not hang the GPU -- its wait polls for the
    configured time, latches `timed_out` (ls_xchg_status) and every later wait returns immediately."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_silent_peer_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = dict((r[0], r[1:]) for r in (q.get(timeout=300) for _ in range(world)))
    [p.join(timeout=60) for p in procs]
    ok, took, timed_out, second = res[0]
    assert ok and res[1][0]
    assert timed_out, "the wait must give up"
    assert took < 20.0, f"gave up only after {took:.1f} s"           # (0.2 s of polls; generous: a poll's period is approximate)
    assert second < 2.0, f"a latched exchange waited again ({second:.1f} s)"
