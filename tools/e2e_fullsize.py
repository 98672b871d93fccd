"""End-to-end run of the public generate API at the benchmark's model size WITH a real prefill (bench.py fills the
caches synthetically): Llama-3-8B dimensions, random weights, a random prompt of --prompt tokens.
Checks the lossless property at scale (tree / chain decoding reproduce vanilla decoding token by token) and reports
prefill and decode times, plus a logit-margin log (teacher-forced target logits behind every vanilla decision) that
attributes any divergence to a measured margin.   python tools/e2e_fullsize.py [--prompt 16384] [--gen 96] [--seed 99]"""
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
    ap.add_argument("--prompt", type=int, default=16384)
    ap.add_argument("--gen", type=int, default=96)
    ap.add_argument("--model", default="llama3-8b-262k")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--seed", type=int, default=99)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = bench.make_config(args.model)
    m = bench.build_model(cfg, dev, 0.02, seed=1234)
    if args.no_graphs:
        m.GRAPH_ROUNDS = False
    g = torch.Generator(device=dev).manual_seed(args.seed)
    ids = torch.randint(5, cfg.vocab_size - 5, (1, args.prompt), generator=g, device=dev)
    pl = torch.tensor([args.prompt], device=dev)
    res = {"model": args.model, "prompt_tokens": args.prompt, "gen": args.gen, "seed": args.seed}

    def timed(fn, *a, **k):
        torch.cuda.synchronize()
        t0 = time.time()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        return r, time.time() - t0

    (v_out, v_num, v_dec), v_wall = timed(m.vanilla_generate, ids, pl, max_gen_len=args.gen, eos_id=-1)
    res["vanilla"] = {"wall_s": round(v_wall, 3), "decode_s": round(v_dec, 3), "prefill_s": round(v_wall - v_dec, 3),
                      "tok_per_s": round((args.gen - 1) / v_dec, 2)}
    (t_out, t_count, t_num, t_dec, _), t_wall = timed(m.tree_spec_generate, ids, pl, max_gen_len=args.gen, eos_id=-1)
    n_t = min(int(t_count) + int(t_num), args.gen)
    res["tree"] = {"wall_s": round(t_wall, 3), "decode_s": round(t_dec, 3), "prefill_s": round(t_wall - t_dec, 3),
                   "rounds": int(t_num), "tokens": n_t, "tau": round(n_t / max(int(t_num), 1), 3), "tok_per_s": round(n_t / t_dec, 2)}
    (s_out, s_count, s_num, s_dec, _), s_wall = timed(m.spec_generate, ids, pl, gamma=4, max_gen_len=args.gen, eos_id=-1)
    n_s = min(int(s_count) + int(s_num), args.gen)
    res["chain"] = {"decode_s": round(s_dec, 3), "rounds": int(s_num), "tokens": n_s, "tok_per_s": round(n_s / s_dec, 2)}

    def agree(a, b, n):
        neq = (a[0, :n] != b[0, :n]).nonzero()
        return n if neq.numel() == 0 else int(neq[0])
    res["tree_equals_vanilla_for"] = agree(t_out, v_out, n_t)
    res["chain_equals_vanilla_for"] = agree(s_out, v_out, n_s)
    res["lossless"] = bool(res["tree_equals_vanilla_for"] == n_t and res["chain_equals_vanilla_for"] == n_s)
    res["tree_equals_chain_for"] = agree(t_out, s_out, min(n_t, n_s))
    # Logit-margin log: one teacher-forced pass of the target over prompt + the vanilla tokens gives the logits behind every
    # vanilla decision; a speculative run may only part from vanilla where the top-1 / top-2 margin is within the rounding
    # noise of the two execution paths (fp16 logits: 1 ulp at 5.5 is 2^-8 = 0.0039).
    n_v = int(v_out.size(1))
    full = torch.cat([ids, v_out[:, :n_v - 1]], dim=1)
    m._set_hints(full.size(1), full.size(1))
    with torch.inference_mode():
        h = m.model.forward(full, exec_type="prefill").last_hidden_state
        lg = m.lm_head(h[:, -n_v:]).float()[0]                  # row i: the logits that chose v_out[i]
    top = lg.topk(3, dim=-1)
    margin = (top.values[:, 0] - top.values[:, 1])
    ulp = torch.tensor([abs(float(x)) for x in top.values[:, 0]]).clamp_min(2.0 ** -14).log2().floor().sub(10).exp2()
    margin_ulps = (margin.cpu() / ulp)
    agree_tf = (top.indices[:, 0] == v_out[0, :n_v]).cpu()
    order = margin.argsort()
    res["margins"] = {
        "positions": n_v,
        "teacher_forced_top1_equals_vanilla": int(agree_tf.sum()),
        "min": round(float(margin.min()), 5), "median": round(float(margin.median()), 5),
        "below_1_ulp": int((margin_ulps < 1).sum()), "below_4_ulp": int((margin_ulps < 4).sum()),
        "five_smallest": [{"pos": int(i), "margin": round(float(margin[i]), 5), "ulps": round(float(margin_ulps[i]), 2)} for i in order[:5]],
    }

    def diff_report(name, out, n):
        k = agree(out, v_out, n)
        if k >= n:
            return None
        rank = int((margin < margin[k]).sum())
        return {"pos": k, "vanilla": int(v_out[0, k]), name: int(out[0, k]),
                "teacher_forced_top3": top.indices[k].tolist(), "top3_logits": [round(x, 4) for x in top.values[k].tolist()],
                "margin": round(float(margin[k]), 5), "margin_ulps": round(float(margin_ulps[k]), 2),
                "margin_rank_among_positions": rank,
                "pair_is_top2": {int(out[0, k]), int(v_out[0, k])} == set(top.indices[k, :2].tolist())}
    for name, out, n in (("tree", t_out, n_t), ("chain", s_out, n_s)):
        d = diff_report(name, out, n)
        if d is not None:
            res[f"first_diff_{name}"] = d
    # every divergence must sit on a margin of at most 2 fp16 ulps BETWEEN the two tokens chosen: anything else is a parity bug
    res["divergences_explained_by_margin"] = all(
        (res.get(f"first_diff_{n}") is None) or (res[f"first_diff_{n}"]["margin_ulps"] <= 2 and res[f"first_diff_{n}"]["pair_is_top2"])
        for n in ("tree", "chain"))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
