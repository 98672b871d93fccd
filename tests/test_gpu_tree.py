"""Parity of the beam-tree bookkeeping kernels (ls_tree_grow / ls_tree_verify_inputs / ls_tree_collapse /
ls_tree_commit / ls_embed_rows, called through the C ABI) against the reference's tensor-op formulation
(tests/oracle_ops.py restates llama_glide.py:1019-1121 line by line).  Integer work: bit exact."""
import numpy as np
import pytest
import torch

import cases
import oracle_ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from longspec_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def g(t):
    return t.to(DEV) if torch.is_tensor(t) else t


def pack_bits(mask):
    """dense [b,M,N] 0/1 -> int32 words [b,M,ceil(N/32)] on the CPU (bit j of word w = column 32 w + j)."""
    b, M, N = mask.shape
    words = (N + 31) // 32
    m = torch.zeros((b, M, words * 32), dtype=torch.int64)
    m[:, :, :N] = (mask != 0).long()
    w = (m.view(b, M, words, 32) << torch.arange(32)).sum(-1)
    return w.to(torch.int32)            # wraps to the same bits


def new_state(b, Fn, first):
    tree_mask = torch.zeros((b, Fn, Fn), dtype=torch.int64)
    tree_mask[:, :, 0] = 1
    all_spec = torch.zeros((b, Fn), dtype=torch.int64)
    all_spec[:, 0] = first
    logp = torch.zeros((b, Fn), dtype=torch.float32)
    return tree_mask, all_spec, logp


def grow_both(ops, shape, b, vocab, seed, base0=1000, a=3):
    """Grow a random tree level by level on the GPU and with the reference's tensor ops; compare everything."""
    rng = np.random.RandomState(seed)
    acc_n = [1]
    for c in shape:
        acc_n.append(acc_n[-1] + c)
    Fn = acc_n[-1]
    cpu = new_state(b, Fn, 7)
    dev = tuple(g(t.clone()) for t in cpu)
    base_c = torch.full((b,), base0, dtype=torch.int32) + torch.arange(b, dtype=torch.int32)
    base_d = g(base_c.clone())
    for ms, k in enumerate(shape):
        lo, mid = (0, 1) if ms == 0 else (acc_n[ms - 1], acc_n[ms])
        fathers = rng.randint(0, mid - lo, size=(b, k))
        toks = rng.randint(0, vocab, size=(b, k))
        idx = torch.from_numpy(fathers * vocab + toks).long()
        if ms == 0:
            idx = torch.from_numpy(toks).long()
        vals = torch.from_numpy(-np.sort(rng.rand(b, k).astype(np.float32), axis=1))
        add = a - 1 if ms == 0 else 0
        want_next = ms + 1 < len(shape)
        pos_c, bits_c = oracle_ops.tree_grow(*cpu, vals, idx, vocab, lo, mid, base=base_c, base_add=add, want_next=want_next)
        pos_d, bits_d = ops.tree_grow(*dev, g(vals), g(idx), vocab, lo, mid, base=base_d, base_add=add, want_next=want_next)
        for tc, td, name in zip(cpu, dev, ("tree_mask", "all_spec", "logp_sum")):
            assert torch.equal(tc, td.cpu()), f"{name} differs at level {ms}"
        assert torch.equal(base_c, base_d.cpu())
        if want_next:
            assert torch.equal(pos_c, pos_d.cpu()), f"positions differ at level {ms}"
            assert torch.equal(pack_bits(bits_c), bits_d.cpu()), f"mask bits differ at level {ms}"
        else:
            assert pos_d is None and bits_d is None
    return cpu, dev, acc_n


@pytest.mark.parametrize("shape,b,vocab", [([4, 16, 16, 16, 16], 1, 128256), ([4, 16, 16, 16, 16], 2, 32000), ([2, 3], 1, 64),
                                          ([1, 1, 1], 3, 152064), ([8, 64, 64], 1, 1000), ([3, 30, 40, 64, 64, 64], 2, 4096)],
                         ids=["default", "b2", "tiny", "chain", "wide", "F266"])
def test_tree_grow_matches_reference_ops(ops, shape, b, vocab):
    grow_both(ops, shape, b, vocab, seed=len(shape) * 7 + b)


@pytest.mark.parametrize("shape,b", [([4, 16, 16, 16, 16], 1), ([2, 3], 2), ([8, 64, 64], 1)], ids=["default", "tiny_b2", "wide"])
def test_tree_verify_inputs(ops, shape, b):
    (tm, spec, _), (tm_d, spec_d, _), acc_n = grow_both(ops, shape, b, 5000, seed=11)
    Fn, gamma = acc_n[-1], len(shape)
    R = Fn - 1 + gamma + 1
    rng = np.random.RandomState(3)
    for a in range(1, gamma + 2):
        acc_pad = torch.from_numpy(rng.randint(1, 5000, size=(b, gamma + 1))).long()
        lens = torch.from_numpy(rng.randint(100, 100000, size=(b,))).int()
        bump_c = torch.full((b,), 77, dtype=torch.int32)
        bump_d = g(bump_c.clone())
        v_c, p_c, m_c = oracle_ops.tree_verify_inputs(acc_pad[:, :a], a, spec, tm, lens, R, bump=bump_c, bump_add=1)
        v_d, p_d, m_d = ops.tree_verify_inputs(g(acc_pad)[:, :a], a, spec_d, tm_d, g(lens), R, bump=bump_d, bump_add=1)
        assert torch.equal(v_c, v_d.cpu()) and torch.equal(p_c, p_d.cpu()), f"a={a}"
        assert torch.equal(pack_bits(m_c), m_d.cpu()), f"a={a}"
        assert torch.equal(bump_c, bump_d.cpu())
        # the packed mask is what ls_pack_tree_mask gives on the dense one
        assert torch.equal(ops.pack_tree_mask(g(m_c)).cpu(), m_d.cpu())


@pytest.mark.parametrize("b,Fn,g1,cap,emitted,eos", [(1, 69, 6, 64, 1, 5), (1, 69, 6, 64, 60, 3), (2, 6, 3, 40, 10, None),
                                                      (1, 137, 4, 3000, 2990, 9), (3, 10, 5, 17, 15, 0)])
def test_tree_commit(ops, b, Fn, g1, cap, emitted, eos):
    rng = np.random.RandomState(b * 100 + Fn)
    for trial in range(6):
        acc_num = torch.from_numpy(rng.randint(1, g1 + 1, size=(b,))).long()
        acc_ids = torch.from_numpy(rng.randint(1, 12, size=(b, g1))).long()
        acc_ids[torch.arange(g1)[None, :] >= acc_num[:, None]] = 0
        out = torch.from_numpy(rng.randint(10, 20, size=(b, cap))).long()
        if trial == 1 and eos is not None:
            out[:, cap - 1] = eos                                   # a stale id anywhere in the buffer counts (:1120)
        tm = torch.from_numpy(rng.randint(0, 2, size=(b, Fn, Fn))).long()
        spec = torch.from_numpy(rng.randint(0, 99, size=(b, Fn))).long()
        logp = torch.from_numpy(rng.randn(b, Fn).astype(np.float32))
        tl = torch.from_numpy(rng.randint(0, 1000, size=(b,))).int()
        dl = torch.from_numpy(rng.randint(0, 1000, size=(b,))).int()
        cpu = [t.clone() for t in (out, tm, spec, logp, tl, dl)]
        dev = [g(t.clone()) for t in (out, tm, spec, logp, tl, dl)]
        st_c = oracle_ops.tree_commit(acc_ids, acc_num, cpu[0], emitted, eos, cpu[1], cpu[2], cpu[3], target_lens=cpu[4],
                                      target_add=4, draft_kv_lens=cpu[5])
        st_d = ops.tree_commit(g(acc_ids), g(acc_num), dev[0], emitted, eos, dev[1], dev[2], dev[3], target_lens=dev[4],
                               target_add=4, draft_kv_lens=dev[5])
        assert torch.equal(st_c, st_d.cpu())
        for tc, td, name in zip(cpu, dev, ("output_ids", "tree_mask", "all_spec", "logp_sum", "target_lens", "draft_kv_lens")):
            assert torch.equal(tc, td.cpu()), name


def test_tree_collapse_len_add_and_weighted_masks(ops):
    """cache_len_add shifts the moved rows; mask entries > 1 keep the argmax / sum semantics of :1136-1144."""
    rng = np.random.RandomState(9)
    import toy
    parents = toy.random_beam_tree([4, 16, 16, 16, 16], 3)
    mask = torch.from_numpy(toy.tree_mask_from_parents(parents))[None].clone()
    mask[0, 5, 5] = 2
    mask[0, 20, 1] = 3
    Fn = mask.shape[1]
    spec = torch.from_numpy(rng.randint(2, 5, size=(1, Fn))).long()
    pred = torch.from_numpy(rng.randint(2, 5, size=(1, Fn))).long()
    kc = torch.from_numpy(rng.randn(1, 400, 2, 128).astype(np.float16))
    vc = torch.from_numpy(rng.randn(1, 400, 2, 128).astype(np.float16))
    lens = torch.tensor([200], dtype=torch.int32)
    kc_d, vc_d = g(kc.clone()), g(vc.clone())
    ids_d, num_d, dbl_d, map_d = ops.tree_collapse(g(spec), g(pred), g(mask), g(lens), 53, 6, kc_d, vc_d, cache_len_add=4)
    ids_c, num_c, dbl_c, map_c = oracle_ops.tree_collapse(spec, pred, mask, lens, 53, 6, kc, vc, cache_len_add=4)
    assert torch.equal(num_c, num_d.cpu()) and torch.equal(dbl_c.to(torch.int32), dbl_d.cpu().to(torch.int32))
    assert torch.equal(ids_c, ids_d.cpu()) and torch.equal(map_c, map_d.cpu())
    assert torch.equal(kc, kc_d.cpu()) and torch.equal(vc, vc_d.cpu())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 1), (1, 16), (2, 37), (1, 74), (128,)])
def test_embed_rows(ops, dtype, shape):
    torch.manual_seed(0)
    w = torch.randn(5000, 4096).to(dtype)
    ids = torch.randint(0, 5000, shape)
    got = ops.embed_rows(g(w), g(ids))
    assert got.shape == (*shape, 4096)
    assert torch.equal(got.cpu(), torch.nn.functional.embedding(ids, w))
    assert ops.embed_supported(g(ids), g(w)) and not ops.embed_supported(g(torch.zeros(1, 200, dtype=torch.long)), g(w))


@pytest.mark.parametrize("b,gamma,eos", [(1, 4, 7), (1, 4, None), (3, 4, 5), (2, 7, 3), (1, 1, 2)])
def test_chain_commit(ops, b, gamma, eos):
    """End of a chain-speculation round (llama_glide.py:738-770) against the reference's tensor ops."""
    rng = np.random.RandomState(b * 10 + gamma)
    for trial in range(12):
        spec = torch.from_numpy(rng.randint(1, 9, size=(b, gamma + 1))).long()
        llm = torch.from_numpy(rng.randint(1, 9, size=(b, gamma + 1))).long()
        n_match = rng.randint(0, gamma + 1, size=b)                     # force every acceptance length to occur
        for z in range(b):
            llm[z, :n_match[z]] = spec[z, 1:1 + n_match[z]]
        input_len = torch.from_numpy(rng.randint(5, 50, size=(b,))).int()
        gen = torch.from_numpy(rng.randint(0, 30, size=(b,))).int()
        cache_lens = input_len + gen
        cap = 30 + 2 * gamma + 4
        out = torch.from_numpy(rng.randint(10, 20, size=(b, cap))).long()
        if trial % 3 == 0 and eos is not None:
            out[:, 0] = eos
        nss = torch.from_numpy(rng.randint(1, 9, size=(b, 2))).long()
        dcl = torch.zeros(b, dtype=torch.int32)
        cpu = [t.clone() for t in (llm, spec, out, cache_lens, dcl, nss)]
        dev = [g(t.clone()) for t in (llm, spec, out, cache_lens, dcl, nss)]
        st_c = oracle_ops.chain_commit(cpu[0], cpu[1], cpu[2], cpu[3], cpu[4], input_len, cpu[5], eos)
        st_d = ops.chain_commit(dev[0], dev[1], dev[2], dev[3], dev[4], g(input_len), dev[5], eos)
        assert torch.equal(st_c, st_d.cpu()), f"trial {trial}"
        for tc, td, name in zip(cpu, dev, ("llm_verify_output", "spec_buffer", "output_ids", "cache_lens", "draft_cache_lens",
                                           "next_spec_start_token")):
            assert torch.equal(tc, td.cpu()), f"{name} (trial {trial})"


def test_largest_tree_all_operators(ops):
    """928 nodes (the operators take up to 1024), 16 levels of up to 64 nodes: growth, verification inputs, collapse and
    commit against the reference's tensor ops."""
    shape = [7, 24] + [64] * 14
    (tm, spec, logp), (tm_d, spec_d, logp_d), acc_n = grow_both(ops, shape, 1, 32000, seed=77)
    Fn, gamma = acc_n[-1], len(shape)
    assert Fn == 928
    R = Fn - 1 + gamma + 1
    rng = np.random.RandomState(5)
    acc_pad = torch.from_numpy(rng.randint(1, 5000, size=(1, gamma + 1))).long()
    lens = torch.tensor([12345], dtype=torch.int32)
    a = 3
    v_c, p_c, m_c = oracle_ops.tree_verify_inputs(acc_pad[:, :a], a, spec, tm, lens, R)
    v_d, p_d, m_d = ops.tree_verify_inputs(g(acc_pad)[:, :a], a, spec_d, tm_d, g(lens), R)
    assert torch.equal(v_c, v_d.cpu()) and torch.equal(p_c, p_d.cpu()) and torch.equal(pack_bits(m_c), m_d.cpu())
    # the target agrees with the draft along one root-to-leaf path and nowhere else
    pred = torch.full((1, Fn), 31999, dtype=torch.int64)
    node = Fn - 1
    while node != 0:
        row = tm[0, node].clone()
        row[node] = 0
        father = int(row.nonzero().max())
        pred[0, father] = spec[0, node]
        node = father
    ids_c, num_c, dbl_c, map_c = oracle_ops.tree_collapse(spec, pred, tm, lens, acc_n[-2], gamma + 1)
    ids_d, num_d, dbl_d, map_d = ops.tree_collapse(spec_d, g(pred), tm_d, g(lens), acc_n[-2], gamma + 1)
    assert int(num_c[0]) == gamma + 1                                  # the whole path is accepted
    assert torch.equal(num_c, num_d.cpu()) and torch.equal(ids_c, ids_d.cpu()) and torch.equal(map_c, map_d.cpu())
    assert int(dbl_c[0]) == int(dbl_d[0])
    out_c = torch.zeros((1, 64), dtype=torch.int64)
    out_d = g(out_c.clone())
    tl = torch.tensor([100], dtype=torch.int32)
    st_c = oracle_ops.tree_commit(ids_c, num_c, out_c, 5, 31999, tm, spec, logp, target_lens=tl.clone(), target_add=a)
    st_d = ops.tree_commit(ids_d, num_d, out_d, 5, 31999, tm_d, spec_d, logp_d, target_lens=g(tl.clone()), target_add=a)
    assert torch.equal(st_c, st_d.cpu()) and torch.equal(out_c, out_d.cpu())
    assert torch.equal(tm, tm_d.cpu()) and torch.equal(spec, spec_d.cpu())


# --------------------------------------------------------------------------- #
# temperature > 0: ls_tree_verify_stochastic
# --------------------------------------------------------------------------- #
def _cpu_noise(V, dtype, device):
    """torch.multinomial(p, 1) on the reference's (CPU) generator = one exponential_ row of p's dtype."""
    return torch.empty(V, dtype=dtype).exponential_(1).to(device)


@pytest.mark.parametrize("c", list(cases.stochastic_cases()), ids=lambda c: c["name"])
def test_verify_stochastic_golden(c):
    """The device walk with the reference's random streams (Python's generator seeded like the golden run, the
    Exponential(1) row of torch.multinomial drawn from the CPU generator) == LlamaGlide.verify_stochastic of the
    reference, token for token; Python's stream ends where the reference left it."""
    import random
    from longspec_amd import ops
    ops.stochastic_noise_fn = _cpu_noise
    try:
        random.seed(5000 + c["ci"])
        torch.manual_seed(6000 + c["ci"])
        acc_ids, acc_num = ops.verify_stochastic(c["spec"].cuda(), c["mask"].cuda(), c["logits"].cuda(), c["spec_logp"].cuda(), c["T"])
        assert torch.equal(acc_num.cpu(), c["acc_num"])
        assert torch.equal(acc_ids.cpu(), c["acc_ids"])
        assert random.random() == c["after_random"]
    finally:
        ops.stochastic_noise_fn = None


def test_verify_stochastic_bf16_and_batch_vs_oracle():
    """bf16 and bsz = 2 against the oracle restatement with the same draws."""
    import random
    from longspec_amd import ops
    from oracle import ref_ops
    ops.stochastic_noise_fn = _cpu_noise
    try:
        ins = [cases.stochastic_inputs(ci) for ci in (1, 5)]
        spec = torch.cat([i[1] for i in ins]); mask = torch.cat([i[2] for i in ins])
        logits = torch.cat([i[3] for i in ins]).to(torch.bfloat16); logp = torch.cat([i[4] for i in ins])
        random.seed(11); torch.manual_seed(12)
        want_ids, want_num = ref_ops.verify_stochastic(spec, mask, logits.clone(), logp.clone(), 0.9)
        after = random.random()
        random.seed(11); torch.manual_seed(12)
        got_ids, got_num = ops.verify_stochastic(spec.cuda(), mask.cuda(), logits.cuda(), logp.cuda(), 0.9)
        assert torch.equal(got_num.cpu(), want_num) and torch.equal(got_ids.cpu(), want_ids)
        assert random.random() == after
    finally:
        ops.stochastic_noise_fn = None
