"""Pin the CPU oracle (``oracle/ref_ops.py``) against golden vectors produced by
the reference itself (``tests/golden/make_golden.py``).  CPU only."""
import numpy as np
import pytest
import torch

import cases
from oracle import ref_ops


def _maxdiff(a, b):
    return (a.float() - b.float()).abs().max().item()


@pytest.mark.parametrize("c", list(cases.triton_cases()), ids=lambda c: c["name"])
def test_triton_tree_kernel_restatement(c):
    """G-a: blocked restatement == the real Triton kernel (interpreter), bit for bit
    on o (fp16) and to 1e-6 on L."""
    o, L = ref_ops.triton_tree_attention(c["q"], c["k"], c["v"], c["mask"])
    assert _maxdiff(L, c["L"]) <= 2e-6
    # identical block order and rounding points; allow one fp16 ulp for exp2 library differences
    d = (o.float() - c["o"].float()).abs()
    assert d.max().item() <= 1e-3
    assert (d > 0).float().mean().item() < 0.02


@pytest.mark.parametrize("c", list(cases.tree_part_cases()), ids=lambda c: c["name"])
def test_target_tree_part(c):
    """G-b: LlamaAttention.tree_part_fwd (current_out, weight) -- bit exact."""
    cur, w = ref_ops.target_tree_part(c["q"], c["k"], c["v"], c["mask"], c["prefix_lse"], c["last_layer"])
    assert torch.equal(cur, c["current_out"])
    assert torch.equal(w, c["weight"])


@pytest.mark.parametrize("c", list(cases.verify_cases()), ids=lambda c: c["name"])
def test_hybrid_verify_attention(c):
    """G-c: the reference's own hybrid tree_decoding (its tree part + fp16 merge over
    the restated flash-attn contract) -- bit exact; and hybrid ~= the reference's dense
    twin tree_decoding_torch to fp16 rounding."""
    for last in (False, True):
        kc, vc = c["kc"].clone(), c["vc"].clone()
        out = ref_ops.target_verify_attention(c["q"], c["k"], c["v"], kc, vc, c["cache_lens"], c["mask"], last)
        assert torch.equal(out, c["hybrid"][last])
        L = c["L"]
        assert torch.equal(kc[:, L:L + 74], c["k"]) and torch.equal(vc[:, L:L + 74], c["v"])
    if c["dense"] is not None:
        out = ref_ops.target_verify_attention(c["q"], c["k"], c["v"], c["kc"].clone(), c["vc"].clone(),
                                              c["cache_lens"], c["mask"], False)
        assert _maxdiff(out, c["dense"]) <= 4e-3          # two fp16 roundings apart on |o| <~ 2
        dense = ref_ops.dense_tree_attention(c["q"], c["k"], c["v"], c["kc"], c["vc"], c["cache_lens"], c["mask"])
        assert _maxdiff(dense, c["dense"]) <= 2e-3


@pytest.mark.parametrize("c", list(cases.draft_cases()), ids=lambda c: c["name"])
def test_draft_attention(c):
    """Draft self-attention seams (GlideAttention.decoding / tree_decoding)."""
    kc, vc = c["kc"].clone(), c["vc"].clone()
    for st in c["steps"]:
        if st["kind"] == "step0":
            out = ref_ops.draft_self_attention_step0(st["q"], st["k"], st["v"], kc, vc, st["cache_lens"])
            assert torch.equal(out, st["out"])
        else:
            out = ref_ops.draft_tree_self_attention(st["q"], st["k"], st["v"], kc, vc, st["cache_lens"], st["mask"])
            d = (out.float() - st["out"].float()).abs()
            assert d.max().item() <= 1e-3 and (d > 0).float().mean().item() < 0.02


@pytest.mark.parametrize("c", list(cases.tree_verification_cases()), ids=lambda c: c["name"])
def test_tree_verification(c):
    """G-d: accept/reject tree collapse incl. the last-layer KV row move -- bit exact."""
    acc_ids, acc_num, dbl, idx_map = ref_ops.tree_verification(c["spec"], c["pred"], c["mask"], c["non_leaf_len"])
    assert torch.equal(acc_ids, c["acc_ids"])
    assert torch.equal(acc_num, c["acc_num"])
    assert torch.equal(dbl.to(c["double_input"].dtype), c["double_input"])
    kc, vc = c["kc"].clone(), c["vc"].clone()
    ref_ops.move_accepted_kv(kc, vc, torch.tensor([c["cache_len"]]), idx_map)
    assert torch.equal(kc, c["kc_after"]) and torch.equal(vc, c["vc_after"])


def test_tree_verification_worked_example():
    """SURVEY 3.4 worked example."""
    import toy
    parents = np.array([0, 0, 0, 1, 2, 3, 3])
    mask = torch.from_numpy(toy.tree_mask_from_parents(parents))[None]
    spec = torch.tensor([[10, 11, 12, 13, 14, 15, 16]])
    pred = torch.tensor([[11, 13, 99, 16, 77, 55, 42]])
    acc_ids, acc_num, dbl, idx = ref_ops.tree_verification(spec, pred, mask, non_leaf_len=5)
    assert acc_ids.tolist() == [[11, 13, 16, 42]] and acc_num.tolist() == [4] and dbl.tolist() == [1]
    assert idx.tolist() == [[0, 1, 3, 6]]


@pytest.mark.parametrize("c", list(cases.norm_cases()), ids=lambda c: c["name"])
def test_rmsnorm(c):
    assert torch.equal(ref_ops.rmsnorm(c["x"], c["w"], c["eps"]), c["y"])


@pytest.mark.parametrize("c", list(cases.rope_cases()), ids=lambda c: c["name"])
def test_rope(c):
    cos, sin = ref_ops.rope_cos_sin(c["pos"], c["inv_freq"], c["scaling"])
    assert torch.equal(cos, c["cos"]) and torch.equal(sin, c["sin"])
    assert torch.equal(ref_ops.apply_rope(c["q"], cos, sin), c["q_out"])
    assert torch.equal(ref_ops.apply_rope(c["k"], cos, sin), c["k_out"])


@pytest.mark.parametrize("c", list(cases.decoding_torch_cases()), ids=lambda c: c["name"])
def test_kvcache_attention_vs_reference_decoding_torch(c):
    """G-h: the restated flash-attn contract against an output the reference ITSELF produced -- its dense
    decode step ``LlamaAttention.decoding_torch`` (longspec/test/llama.py:161-197), run by tests/golden/make_golden.py with no
    stub of ours on the path.  The twin rounds differently from a flash kernel (fp16 QK^T product, fp16 probabilities over
    the whole row), so the bound is two fp16 roundings at |o| <~ 1, not bit equality; the appended cache rows are exact."""
    kc, vc = c["kc"].clone(), c["vc"].clone()
    L, a = c["L"], c["a"]
    out = ref_ops.kvcache_attention(c["q"], kc, vc, c["k"], c["v"], cache_seqlens=torch.tensor([L], dtype=torch.int32), causal=True)
    assert torch.equal(kc[:, L:L + a], c["k_rows"]) and torch.equal(vc[:, L:L + a], c["v_rows"])
    d = (out.float() - c["out"].float()).abs()
    print(f"G-h {c['name']}: max|diff| {d.max().item():.3e} mean {d.mean().item():.3e}")
    assert d.max().item() <= 1.1e-3 and d.mean().item() <= 1.7e-4      # observed 7.3e-4 / 1.1e-4; 1 fp16 ulp at |o| ~ 1 is 9.8e-4


def test_kvcache_attention_matches_dense_decoding():
    """The restated flash-attn contract against the reference's dense decode twin
    (decoding_torch, longspec/test/llama.py:183-192) semantics: causal append."""
    import toy
    H, Hkv, L, a = 4, 2, 50, 3
    q = toy.randn_f16((1, a, H, 128), 1)
    k = toy.randn_f16((1, a, Hkv, 128), 2)
    v = toy.randn_f16((1, a, Hkv, 128), 3)
    kc = torch.zeros(1, L + 8, Hkv, 128, dtype=torch.float16)
    vc = torch.zeros(1, L + 8, Hkv, 128, dtype=torch.float16)
    kc[:, :L] = toy.randn_f16((1, L, Hkv, 128), 4)
    vc[:, :L] = toy.randn_f16((1, L, Hkv, 128), 5)
    out = ref_ops.kvcache_attention(q, kc, vc, k, v, cache_seqlens=torch.tensor([L], dtype=torch.int32), causal=True)
    K = kc[:, :L + a].repeat_interleave(H // Hkv, dim=2).permute(0, 2, 3, 1).float()
    V = vc[:, :L + a].repeat_interleave(H // Hkv, dim=2).transpose(1, 2).float()
    s = torch.matmul(q.transpose(1, 2).float(), K) / (128 ** 0.5)
    msk = torch.cat((torch.zeros(a, L, dtype=torch.bool), torch.triu(torch.ones(a, a), diagonal=1).bool()), dim=1)
    s = s.masked_fill(msk, float("-inf"))
    ref = torch.matmul(torch.softmax(s, -1), V).transpose(1, 2)
    assert _maxdiff(out, ref) <= 2e-3


def test_window_semantics_g3():
    """Gotcha G3: non-causal window (512,-1) without new keys -- row i of sq sees keys
    j >= p - sq + i - 512."""
    vis = ref_ops._bottom_right_mask(16, 700, False, (512, -1))
    for i in (0, 7, 15):
        lo = 700 - 16 + i - 512
        assert not vis[i, lo - 1] and vis[i, lo] and vis[i, 699]
    vis = ref_ops._bottom_right_mask(3, 703, True, (512, -1))     # step 0: self + 512 previous
    for i in range(3):
        assert vis[i].sum().item() == 513 and vis[i, 700 + i] and not vis[i, min(702, 700 + i + 1)] or i == 2


def test_lse_merge_equals_joint_softmax():
    import toy
    H, R, D = 4, 5, 128
    q = toy.randn_f16((1, R, H, D), 1)
    kc = toy.randn_f16((1, 96, H, D), 2)
    vc = toy.randn_f16((1, 96, H, D), 3)
    full, lse_full = ref_ops.kvcache_attention(q, kc, vc, cache_seqlens=torch.tensor([96]), return_softmax_lse=True)
    parts, lses = [], []
    for lo, hi in ((0, 40), (40, 41), (41, 96)):
        o, l = ref_ops.kvcache_attention(q, kc[:, lo:hi].contiguous(), vc[:, lo:hi].contiguous(),
                                         cache_seqlens=torch.tensor([hi - lo]), return_softmax_lse=True)
        parts.append(o[0].float())
        lses.append(l[0])
    o, lse = ref_ops.lse_merge(parts, lses)
    assert _maxdiff(o, full[0]) <= 2e-3 and _maxdiff(lse, lse_full[0]) <= 1e-5


@pytest.mark.parametrize("c", list(cases.stochastic_cases()), ids=lambda c: c["name"])
def test_verify_stochastic_matches_reference(c):
    """temperature > 0: the restatement replays the reference's accept / reject walk token for token when
    Python's and torch's generators are seeded like the golden run -- and leaves the Python stream where the reference
    left it (same number of draws)."""
    import random
    random.seed(5000 + c["ci"])
    torch.manual_seed(6000 + c["ci"])
    acc_ids, acc_num = ref_ops.verify_stochastic(c["spec"], c["mask"], c["logits"].clone(), c["spec_logp"].clone(), c["T"])
    assert torch.equal(acc_num, c["acc_num"])
    assert torch.equal(acc_ids, c["acc_ids"])
    assert random.random() == c["after_random"]
