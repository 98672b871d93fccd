"""Host logic of longspec_amd (model wiring, cache-length state machine, beam-tree growth,
generate loops) driven on CPU by the oracle's operators, against the reference's end-to-end
golden traces (G-e / G-f): token ids, per-round trees, predictions and acceptance -- exact."""
import pytest
import torch

import cases
import oracle_ops


def build(run):
    from longspec_amd.llama_glide import LlamaGlide
    from longspec_amd.qwen2_glide import Qwen2Glide
    m = (Qwen2Glide if run["family"] == "qwen2" else LlamaGlide)(run["cfg"], ops=oracle_ops, dtype=run.get("dtype", torch.float16))
    missing, unexpected = m.load_state_dict({**run["target_sd"], **{"glide." + k: v for k, v in run["draft_sd"].items()}},
                                            strict=True)
    return m


RUNS = list(cases.generate_runs()) + list(cases.generate_runs("qwen2"))
# bfloat16 (how inference_qwq.py runs QwQ).  The reference's Triton tree kernel cannot be run in bf16 in the build
# container (the interpreter computes in numpy, which has no bfloat16), so the golden generator routes that one seam to
# the reference's own pure-torch twin (GlideAttention.tree_part_fwd): vanilla, chain AND tree runs are valid goldens.
RUNS_BF16 = list(cases.generate_runs("qwen2_bf16"))


@pytest.mark.parametrize("run", RUNS, ids=lambda r: r["name"])
def test_vanilla_generate_matches_reference(run):
    m = build(run)
    out, num, _ = m.vanilla_generate(run["prompt"], torch.tensor([run["prompt_len"]]), max_gen_len=run["max_gen_len"],
                                     eos_id=run["eos_id"])
    assert torch.equal(out, run["vanilla_out"])
    assert num == run["vanilla_num"]


@pytest.mark.parametrize("run", RUNS, ids=lambda r: r["name"])
def test_tree_spec_generate_matches_reference(run):
    m = build(run)
    trace = {"mask": [], "spec": [], "pred": [], "acc": [], "n": []}
    orig = oracle_ops.tree_collapse

    class Spy:
        def __getattr__(self, name):
            return getattr(oracle_ops, name)

        @staticmethod
        def tree_collapse(all_spec, all_llm_pred, tree_mask, cache_lens, non_leaf_len, max_acc, k_cache=None, v_cache=None,
                          cache_len_add=0, out_acc_ids=None):
            trace["mask"].append(tree_mask.clone())
            trace["spec"].append(all_spec.clone())
            trace["pred"].append(all_llm_pred.clone())
            r = orig(all_spec, all_llm_pred, tree_mask, cache_lens, non_leaf_len, max_acc, k_cache, v_cache, cache_len_add, out_acc_ids)
            trace["acc"].append(r[0].clone())
            trace["n"].append(r[1].clone())
            return r

    m.ops = Spy()
    out, count, num, _, _ = m.tree_spec_generate(run["prompt"], torch.tensor([run["prompt_len"]]),
                                                 tree_shape=run["tree_shape"], max_gen_len=run["max_gen_len"],
                                                 eos_id=run["eos_id"])
    assert torch.equal(out, run["tree_out"])
    assert (int(count), int(num)) == (run["tree_count"], run["tree_num"])
    # per-round traces: the draft trees, the target's predictions and the acceptance counts.  A plain
    # random draft ("rand") has near-tied beam candidates deep in the tree, where the 1-ulp noise between
    # the Triton kernel (interpreter) and its restatement can flip a top-k pick: allow <= 10 % of rounds
    # to differ there (the emitted tokens above are exact regardless); every other fixture is exact.
    masks = torch.cat(trace["mask"], 0).to(torch.int8)
    specs = torch.cat(trace["spec"], 0)
    assert masks.shape == run["tr_tree_mask"].shape
    same = [torch.equal(masks[i], run["tr_tree_mask"][i]) and torch.equal(specs[i], run["tr_all_spec"][i])
            for i in range(masks.shape[0])]
    if run["name"] in ("rand", "qwen_rand"):
        assert sum(same) >= 0.9 * len(same)
    else:
        assert all(same)
        assert torch.equal(torch.cat(trace["pred"], 0), run["tr_llm_pred"])
    assert torch.equal(torch.cat(trace["n"], 0), run["tr_acc_num"])
    # lossless: identical to the vanilla continuation
    n_tok = int(count) + int(num)
    if run["eos_id"] in run["vanilla_out"][0].tolist():            # stopped on eos: equal up to and including it
        n_tok = min(n_tok, run["vanilla_out"][0].tolist().index(run["eos_id"]) + 1)
    assert torch.equal(out[0, :n_tok], run["vanilla_out"][0, :n_tok])


@pytest.mark.parametrize("run", RUNS, ids=lambda r: r["name"])
def test_chain_spec_generate_matches_reference(run):
    m = build(run)
    out, count, num, _, _ = m.spec_generate(run["prompt"], torch.tensor([run["prompt_len"]]), gamma=4,
                                            max_gen_len=run["max_gen_len"], eos_id=run["eos_id"])
    assert (int(count), int(num)) == (run["chain_count"], run["chain_num"])
    n = min(int(count) + int(num), run["max_gen_len"])
    assert torch.equal(out[:, :n], run["chain_out"][:, :n])


# The regime every BASELINE configuration runs in: prompt >= 700 tokens -- the draft's 512-row window
# (llama_glide.py:262,300) truncates from round 1, so draft_cache_lens / target_cache_lens_for_draft (:1027,1076,1104) are
# compared with the reference INSIDE the loop -- and >= 64 rounds; three seeds per weight kind, Llama and Qwen2 twins, bf16.
RUNS_LONG = (list(cases.generate_runs("llama_long")) + list(cases.generate_runs("qwen2_long"))
             + list(cases.generate_runs("qwen2_bf16_long")))


def _acc_trace_spy(m, trace):
    orig = m.ops.tree_collapse

    class Spy:
        def __getattr__(self, name):
            return getattr(oracle_ops, name)

        @staticmethod
        def tree_collapse(*a, **k):
            r = orig(*a, **k)
            trace.append(r[1].clone())
            return r

    m.ops = Spy()


@pytest.mark.parametrize("run", RUNS_LONG, ids=lambda r: r["name"])
def test_long_runs_match_reference(run):
    """Token ids, `count`, `num` and the per-round acceptance trace of tree decoding, and ids / counters of chain decoding, on
    the long runs -- exact.  (A bug in the window bookkeeping cannot change token ids -- verification is lossless -- but it
    lowers the acceptance: `num` and the per-round `acc_num` trace are what pin it.)"""
    m = build(run)
    pl = torch.tensor([run["prompt_len"]])
    kw = dict(max_gen_len=run["max_gen_len"], eos_id=run["eos_id"])
    assert run["prompt_len"] >= 700 and run["tree_num"] >= 64
    trace = []
    _acc_trace_spy(m, trace)
    t_out, t_count, t_num, _, _ = m.tree_spec_generate(run["prompt"], pl, tree_shape=run["tree_shape"], **kw)
    assert (int(t_count), int(t_num)) == (run["tree_count"], run["tree_num"])
    assert torch.equal(torch.cat(trace, 0), run["tr_acc_num"])
    assert torch.equal(t_out, run["tree_out"])
    n = int(t_count) + int(t_num)
    assert torch.equal(t_out[0, :n], run["vanilla_out"][0, :n])          # lossless
    m.ops = oracle_ops
    s_out, s_count, s_num, _, _ = m.spec_generate(run["prompt"], pl, gamma=4, **kw)
    assert (int(s_count), int(s_num)) == (run["chain_count"], run["chain_num"])
    n = min(int(s_count) + int(s_num), run["max_gen_len"])
    assert torch.equal(s_out[:, :n], run["chain_out"][:, :n])


@pytest.mark.parametrize("run", list(cases.stochastic_runs(long=True)), ids=lambda r: r["name"])
def test_long_tree_run_with_temperature_matches_reference(run):
    """tree_spec_generate(temperature = 0.8) through a truncating draft window for >= 64 rounds: the reference's seeded run."""
    import random
    m = build(run)
    trace = {"ids": [], "num": []}
    orig = m.verify_stochastic

    def spy(*a, **k):
        r = orig(*a, **k)
        pad = torch.full((1, 8), -1, dtype=torch.int64)
        pad[:, :r[0].shape[1]] = r[0]
        trace["ids"].append(pad)
        trace["num"].append(r[1].clone())
        return r

    m.verify_stochastic = spy
    random.seed(7000 + run["wseed"])
    torch.manual_seed(8000 + run["wseed"])
    out, count, num, _, _ = m.tree_spec_generate(run["prompt"], torch.tensor([run["prompt_len"]]), tree_shape=run["tree_shape"],
                                                 max_gen_len=run["max_gen_len"], temperature=run["temperature"])
    assert torch.equal(torch.cat(trace["num"], 0), run["tr_acc_num"])
    assert torch.equal(torch.cat(trace["ids"], 0), run["tr_acc_ids"])
    assert (int(count), int(num)) == (run["count"], run["num"])
    assert torch.equal(out, run["out"])


BASELINES = list(cases.baseline_runs())


@pytest.mark.parametrize("run", BASELINES, ids=lambda r: r["name"])
def test_magicdec_and_vanilla_torch_match_reference(run):
    """--method magicdec (self-speculation over a StreamingLLM cache) and --method vanilla_torch."""
    m = build(run)
    pl = torch.tensor([run["prompt_len"]])
    out, count, num, _, _ = m.magicdec_generate(run["prompt"], pl, gamma=run["gamma"], max_gen_len=run["max_gen_len"])
    assert (int(count), int(num)) == (run["magicdec_count"], run["magicdec_num"])
    assert torch.equal(out, run["magicdec_out"])
    vt, vnum, _ = m.vanilla_torch_generate(run["prompt"], pl, max_gen_len=run["max_gen_len"])
    assert torch.equal(vt, run["vanilla_torch_out"]) and vnum == run["vanilla_torch_num"]


@pytest.mark.parametrize("run", RUNS_BF16, ids=lambda r: r["name"])
def test_bf16_generate_matches_reference(run):
    m = build(run)
    pl = torch.tensor([run["prompt_len"]])
    kw = dict(max_gen_len=run["max_gen_len"], eos_id=run["eos_id"])
    v_out, v_num, _ = m.vanilla_generate(run["prompt"], pl, **kw)
    assert torch.equal(v_out, run["vanilla_out"]) and v_num == run["vanilla_num"]
    s_out, s_count, s_num, _, _ = m.spec_generate(run["prompt"], pl, gamma=4, **kw)
    assert torch.equal(s_out, run["chain_out"]) and (int(s_count), int(s_num)) == (run["chain_count"], run["chain_num"])
    # tree runs: the reference's bf16 goldens take its pure-torch twin of the Triton tree kernel (GlideAttention.tree_part_fwd,
    # qwen2_glide.py:331-359 -- the interpreter has no bfloat16); token ids, count and num are exact
    t_out, t_count, t_num, _, _ = m.tree_spec_generate(run["prompt"], pl, tree_shape=run["tree_shape"], **kw)
    assert torch.equal(t_out, run["tree_out"])
    assert (int(t_count), int(t_num)) == (run["tree_count"], run["tree_num"])
    n = int(t_count) + int(t_num)
    assert torch.equal(t_out[0, :n], run["vanilla_out"][0, :n])          # lossless


@pytest.mark.parametrize("run", list(cases.stochastic_runs()), ids=lambda r: r["name"])
def test_tree_spec_generate_with_temperature_matches_reference(run):
    """temperature > 0 end to end: the host loop around verify_stochastic -- including the reference's own
    T > 0 bookkeeping (no KV compaction, cache_lens without the +1, whole padded acc_ids rows written to output_ids) --
    replays the reference's seeded run token for token: output_ids, count, num and every round's (acc_ids, acc_num)."""
    import random
    m = build(run)
    trace = {"ids": [], "num": []}
    orig = m.verify_stochastic

    def spy(*a, **k):
        r = orig(*a, **k)
        pad = torch.full((1, 8), -1, dtype=torch.int64)
        pad[:, :r[0].shape[1]] = r[0]
        trace["ids"].append(pad)
        trace["num"].append(r[1].clone())
        return r

    m.verify_stochastic = spy
    random.seed(7000 + run["wseed"])
    torch.manual_seed(8000 + run["wseed"])
    out, count, num, _, _ = m.tree_spec_generate(run["prompt"], torch.tensor([run["prompt_len"]]), tree_shape=run["tree_shape"],
                                                 max_gen_len=run["max_gen_len"], temperature=run["temperature"])
    assert torch.equal(torch.cat(trace["num"], 0), run["tr_acc_num"])
    assert torch.equal(torch.cat(trace["ids"], 0), run["tr_acc_ids"])
    assert (int(count), int(num)) == (run["count"], run["num"])
    assert torch.equal(out, run["out"])


@pytest.mark.parametrize("run", list(cases.chain_stochastic_runs()), ids=lambda r: r["name"])
def test_spec_generate_with_temperature_matches_reference(run):
    """The chain method at temperature > 0 (llama_glide.py:715-736): greedy draft chain, rejection test min(1, p/q) against
    one rand_like per round, resampling from the target's distribution -- the reference's seeded run token for token.  The
    host loop draws from torch's global generator in the reference's order (oracle/ref_ops.py::chain_accept_stochastic)."""
    m = build(run)
    torch.manual_seed(run["torch_seed"])
    fn = m.magicdec_generate if run["method"] == "magicdec" else m.spec_generate
    out, count, num, _, _ = fn(run["prompt"], torch.tensor([run["prompt_len"]]), gamma=4,
                                            max_gen_len=run["max_gen_len"], temperature=run["temperature"])
    assert (int(count), int(num)) == (run["count"], run["num"])
    assert torch.equal(out, run["out"])


def test_trace_compare_accepts_the_oracle_run_without_near_ties():
    """tests/trace_compare.py (the GPU long-run tests' "exact up to explained near-ties" comparison) on the CPU path: the host
    logic on the oracle's operators reproduces every round's draft tree, target predictions and acceptance of the reference's
    long run, so the comparison must find equal trees in every round -- and must reject a run whose tree was tampered with."""
    import trace_compare
    run = [r for r in RUNS_LONG if r["name"] == "long_gqa_s1"][0]
    m = build(run)
    spy = trace_compare.RoundSpy(oracle_ops)
    m.ops = spy
    m.tree_spec_generate(run["prompt"], torch.tensor([run["prompt_len"]]), tree_shape=run["tree_shape"],
                         max_gen_len=run["max_gen_len"], eos_id=run["eos_id"])
    st = trace_compare.compare(spy.rounds, run, run["cfg"].vocab_size, tol=0.02)
    assert st["rounds_compared"] == run["tree_num"] - 1 and st["rounds_with_equal_trees"] == st["rounds_compared"], st
    # a candidate that is NOT a near-tie must be rejected: replace one deep node's token in round 3
    bad = [dict(r) for r in spy.rounds]
    bad[3]["spec"] = bad[3]["spec"].clone()
    bad[3]["spec"][40] = (int(bad[3]["spec"][40]) + 7) % 500 + 2
    with pytest.raises(AssertionError):
        trace_compare.compare(bad, run, run["cfg"].vocab_size, tol=0.02)
