#!/usr/bin/env python3
"""Where, and on what margin, does a GPU run of a long golden generation leave the reference's per-round trace?
Replays tests/golden/generate_long*.npz run NAME on the HIP kernels (eager rounds), spying on the beam growth
(`logprob_topk`, asked for 3 candidates more than the round uses) and on `tree_collapse` (draft tree, target predictions,
acceptance); prints the first round whose draft tree / predictions / acceptance differ from the reference's and the log-prob
margins of the level where the trees part.     python tools/dbg_long_run.py long_qwen_g5_s0 [family]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402


def main():
    name = sys.argv[1]
    fams = [sys.argv[2]] if len(sys.argv) > 2 else ["llama_long", "qwen2_long", "qwen2_bf16_long"]
    run = None
    for f in fams:
        for r in cases.generate_runs(f):
            if r["name"] == name:
                run = r
    from longspec_amd import ops as hip_ops
    from longspec_amd.llama_glide import LlamaGlide
    from longspec_amd.qwen2_glide import Qwen2Glide
    m = (Qwen2Glide if run["family"] == "qwen2" else LlamaGlide)(run["cfg"], device="cuda", dtype=run.get("dtype", torch.float16))
    m.load_state_dict({**run["target_sd"], **{"glide." + k: v for k, v in run["draft_sd"].items()}}, strict=True)
    m.GRAPH_ROUNDS = False
    rounds, cur = [], {"topk": []}

    class Spy:
        def __getattr__(self, n):
            return getattr(hip_ops, n)

        @staticmethod
        def logprob_topk(logits, history, k):
            v, i = hip_ops.logprob_topk(logits, history, k + 3)
            cur["topk"].append((v[0].cpu().tolist(), i[0].cpu().tolist(), k))
            return v[:, :k].contiguous(), i[:, :k].contiguous()

        @staticmethod
        def tree_collapse(all_spec, all_llm_pred, tree_mask, *a, **k):
            cur.update(spec=all_spec.cpu().clone(), pred=all_llm_pred.cpu().clone(), mask=tree_mask.cpu().clone().to(torch.int8))
            r = hip_ops.tree_collapse(all_spec, all_llm_pred, tree_mask, *a, **k)
            cur["acc"] = int(r[1][0])
            rounds.append(dict(cur))
            cur.clear()
            cur["topk"] = []
            return r

    m.ops = Spy()
    ids = run["prompt"].cuda()
    pl = torch.tensor([run["prompt_len"]], device="cuda")
    out, count, num, _, _ = m.tree_spec_generate(ids, pl, tree_shape=run["tree_shape"], max_gen_len=run["max_gen_len"], eos_id=run["eos_id"])
    res = {"run": name, "count_num": [int(count), int(num)], "reference": [run["tree_count"], run["tree_num"]],
           "tokens_equal": bool(torch.equal(out.cpu(), run["tree_out"]))}
    g_spec, g_mask, g_pred, g_acc = run["tr_all_spec"], run["tr_tree_mask"], run["tr_llm_pred"], run["tr_acc_num"]
    V = run["cfg"].vocab_size
    acc_n = [1]
    for c in run["tree_shape"]:
        acc_n.append(acc_n[-1] + c)
    for r, rd in enumerate(rounds):
        if r >= g_spec.shape[0]:
            break
        same_tree = torch.equal(rd["spec"][0], g_spec[r]) and torch.equal(rd["mask"][0], g_mask[r])
        same_pred = torch.equal(rd["pred"][0], g_pred[r])
        if same_tree and same_pred and rd["acc"] == int(g_acc[r]):
            continue
        res["first_round_that_differs"] = r
        res["tree_equal"], res["pred_equal"] = same_tree, same_pred
        res["acc_num"] = [rd["acc"], int(g_acc[r])]
        if not same_tree:
            node = int((rd["spec"][0] != g_spec[r]).nonzero()[0]) if not torch.equal(rd["spec"][0], g_spec[r]) else \
                int((rd["mask"][0] != g_mask[r]).any(dim=-1).nonzero()[0])
            lvl = max(i for i in range(len(acc_n)) if acc_n[i] <= node)
            res["first_node_that_differs"] = {"node": node, "level": lvl, "ours": int(rd["spec"][0, node]), "reference": int(g_spec[r, node])}
            vals, idx, k = rd["topk"][lvl]
            res["level_topk_logprob_sums"] = [round(x, 5) for x in vals]
            res["level_topk_tokens"] = [int(i) % V for i in idx]
            res["level_topk_parents"] = [int(i) // V for i in idx]
            res["k_used"] = k
            res["margin_kth_to_next"] = round(vals[k - 1] - vals[k], 6)
            res["smallest_adjacent_margin_in_topk"] = round(min(vals[i] - vals[i + 1] for i in range(k)), 6)
            res["reference_level_tokens"] = g_spec[r, acc_n[lvl]:acc_n[lvl + 1]].tolist()
            res["our_level_tokens"] = rd["spec"][0, acc_n[lvl]:acc_n[lvl + 1]].tolist()
        break
    else:
        res["first_round_that_differs"] = None
    print(json.dumps(res))


if __name__ == "__main__":
    main()
