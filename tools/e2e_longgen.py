"""BASELINE configs[4] as it is written -- "QwQ-32B-Preview + longspec draft, 20k-token long-CoT generation, 32k prefix, bf16" --
through the public API with a REAL prefill (reference caller: longspec/test/inference_qwq.py:101-136): one tree_spec_generate
and one vanilla_generate of --gen tokens from the same random prompt.  Reports tokens/s over the WHOLE generation (the
reference's (count + num) / elapsed), tau per 1000 emitted tokens, the KV length at start and end, HIP-graph tiers / captures
(LlamaGlide.GRAPH_TIER), peak device memory, the longest common prefix of the two runs and the target's top-1 / top-2 margin
at the first divergence (teacher-forced: one more prefill over prompt + common prefix).

    python tools/e2e_longgen.py [--model qwq-32b] [--prompt 32768] [--gen 20000] [--agreement 0.01]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwq-32b")
    ap.add_argument("--prompt", type=int, default=32768)
    ap.add_argument("--gen", type=int, default=20000)
    ap.add_argument("--agreement", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=99)
    ap.add_argument("--no-vanilla", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = bench.make_config(args.model)
    t0 = time.time()
    m = bench.build_model(cfg, dev, args.agreement, seed=1234)
    g = torch.Generator(device=dev).manual_seed(args.seed)
    ids = torch.randint(5, cfg.vocab_size - 5, (1, args.prompt), generator=g, device=dev)
    pl = torch.tensor([args.prompt], device=dev)
    res = {"model": args.model, "dtype": cfg.dtype, "prompt_tokens": args.prompt, "gen": args.gen, "agreement": args.agreement,
           "workload": bench.BASELINE_CONFIGS[4]["name"] if args.model == "qwq-32b" else args.model,
           "build_s": round(time.time() - t0, 1), "graph_tier": m.GRAPH_TIER, "graph_after": m.GRAPH_AFTER}

    # ---- tree decoding, with a per-round log (the round's one host read has already happened when tree_round returns)
    log, states = [], []
    orig_round, orig_begin = m.tree_round, m.begin_tree_decode

    def begin(*a, **k):
        st = orig_begin(*a, **k)
        states.append(st)
        return st

    def rnd(st):
        r = orig_round(st)
        log.append((st.emitted, time.time()))
        return r

    m.begin_tree_decode, m.tree_round = begin, rnd
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize()
    t0 = time.time()
    t_out, t_count, t_num, t_dec, _ = m.tree_spec_generate(ids, pl, max_gen_len=args.gen, eos_id=-1)
    torch.cuda.synchronize()
    t_wall = time.time() - t0
    m.begin_tree_decode, m.tree_round = orig_begin, orig_round
    st = states[0]
    n_t = min(int(t_count) + int(t_num), args.gen)
    free_b, total_b = torch.cuda.mem_get_info()
    res["tree"] = {"prefill_s": round(t_wall - t_dec, 2), "decode_s": round(t_dec, 2), "rounds": int(t_num), "tokens": n_t,
                   "tau": round(n_t / max(int(t_num), 1), 3), "tok_per_s": round(n_t / t_dec, 2),
                   "ms_per_round": round(1e3 * t_dec / max(int(t_num), 1), 3),
                   "kv_rows_start": args.prompt, "kv_rows_end": args.prompt + n_t,
                   "graph_tiers": st.graph_tiers, "graph_captures": st.graph_captures,
                   "graphs_on": bool(st.use_graphs and st.graphs is not False),
                   "peak_torch_alloc_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                   "device_in_use_GB": round((total_b - free_b) / 2 ** 30, 1)}
    # tau and tokens/s per 1000 emitted tokens
    per_k, e0, t_prev, r0 = [], 1, None, 0
    t_start = log[0][1] - (log[1][1] - log[0][1]) if len(log) > 1 else 0.0
    t_prev = t_start
    nxt = 1000
    for i, (e, t) in enumerate(log):
        if e >= nxt or i == len(log) - 1:
            per_k.append({"upto": e, "tau": round((e - e0) / max(i + 1 - r0, 1), 3),
                          "tok_per_s": round((e - e0) / max(t - t_prev, 1e-9), 1),
                          "ms_per_round": round(1e3 * (t - t_prev) / max(i + 1 - r0, 1), 3)})
            e0, r0, t_prev = e, i + 1, t
            nxt = (e // 1000 + 1) * 1000
    res["tree"]["per_1000_tokens"] = per_k

    if not args.no_vanilla:
        torch.cuda.synchronize()
        t0 = time.time()
        v_out, v_num, v_dec = m.vanilla_generate(ids, pl, max_gen_len=args.gen, eos_id=-1)
        torch.cuda.synchronize()
        v_wall = time.time() - t0
        res["vanilla"] = {"prefill_s": round(v_wall - v_dec, 2), "decode_s": round(v_dec, 2),
                          "tok_per_s": round((args.gen - 1) / v_dec, 2)}
        res["speedup_over_vanilla"] = round(res["tree"]["tok_per_s"] / res["vanilla"]["tok_per_s"], 3)
        n = min(n_t, args.gen)
        neq = (t_out[0, :n] != v_out[0, :n]).nonzero()
        k = n if neq.numel() == 0 else int(neq[0])
        res["tree_equals_vanilla_for"] = k
        if k < n:
            # the logits behind decision k: teacher-forced over prompt + the k common tokens
            full = torch.cat([ids, v_out[:, :k]], dim=1)
            m._clear_shard()
            m.set_max_gen_len(64)
            m._set_hints(full.size(1), full.size(1))
            with torch.inference_mode():
                h = m.model.forward(full, exec_type="prefill").last_hidden_state
                lg = m.lm_head(h[:, -1:]).float()[0, 0]
            top = lg.topk(3)
            ulp = 2.0 ** (torch.tensor(abs(float(top.values[0]))).clamp_min(2.0 ** -14).log2().floor().item() - (7 if cfg.dtype == "bf16" else 10))
            tv, vv = int(t_out[0, k]), int(v_out[0, k])
            res["first_divergence"] = {"pos": k, "tree": tv, "vanilla": vv, "teacher_forced_top3": top.indices.tolist(),
                                       "top3_logits": [round(x, 4) for x in top.values.tolist()],
                                       "margin_top1_top2": round(float(top.values[0] - top.values[1]), 5),
                                       "margin_between_the_two_tokens": round(abs(float(lg[tv] - lg[vv])), 5),
                                       "ulp_of_the_logit_dtype_there": ulp,
                                       "pair_is_top2": {tv, vv} == set(top.indices[:2].tolist())}
            res["divergence_explained_by_margin"] = bool(res["first_divergence"]["margin_between_the_two_tokens"] <= 2 * ulp
                                                         and res["first_divergence"]["pair_is_top2"])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
