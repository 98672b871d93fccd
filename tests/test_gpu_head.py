"""ls_pass_head (round 6): embedding gather + RoPE table + first input RMSNorm of a decode pass in one launch.  It shares its
device code with ls_embed_rows / ls_rope_cos_sin / ls_rmsnorm_fwd, which are pinned to the reference's goldens
(tests/test_gpu_ops.py: norm_rope fixtures) -- so the bar here is BIT-IDENTITY with those three operators, for every row count,
hidden size and dtype a decode pass of the BASELINE models has, with positions given (tree levels, verify pass) or formed from
the cache lengths (draft step 0, vanilla step: `arange + cache_lens[:, None]`, llama_glide.py:1005 / llama.py:571-577)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from longspec_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def _inv_freq(theta=500000.0):
    return (1.0 / (theta ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))).to(DEV)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hidden", [256, 1000, 4096, 5120, 8192, 16384])
@pytest.mark.parametrize("b,q", [(1, 1), (1, 4), (1, 16), (1, 74), (2, 5), (1, 128)])
def test_pass_head_equals_the_three_operators(ops, dtype, hidden, b, q):
    g = torch.Generator().manual_seed(hidden * 131 + b * 17 + q)
    vocab = 3001
    table = (torch.randn(vocab, hidden, generator=g) * 0.7).to(dtype).to(DEV)
    wgt = (1.0 + 0.1 * torch.randn(hidden, generator=g)).to(dtype).to(DEV)
    ids = torch.randint(0, vocab, (b, q), generator=g).to(DEV)
    inv = _inv_freq()
    eps, scaling = 1e-5, 0.8366
    # (a) explicit positions (tree depth order: not monotone)
    pos = torch.randint(0, 140000, (b, q), generator=g).to(DEV)
    emb, nrm, (cos, sin) = ops.pass_head(table, ids, inv, scaling, wgt, eps, position_ids=pos)
    emb_r = ops.embed_rows(table, ids)
    cos_r, sin_r = ops.rope_cos_sin(pos, inv, scaling, dtype)
    nrm_r = ops.rmsnorm(emb_r, wgt, eps)
    assert emb.shape == (b, q, hidden) and nrm.shape == (b, q, hidden) and cos.shape == (b, q, 128)
    assert torch.equal(emb, emb_r) and torch.equal(nrm, nrm_r)
    assert torch.equal(cos, cos_r) and torch.equal(sin, sin_r)
    # (b) positions = arange(q) + base[:, None] + add
    base = torch.tensor([131072 + 7 * i for i in range(b)], dtype=torch.int32, device=DEV)
    emb2, nrm2, (cos2, sin2) = ops.pass_head(table, ids, inv, scaling, wgt, eps, pos_base=base, pos_add=3)
    pos2 = torch.arange(q, device=DEV)[None, :] + base[:, None].long() + 3
    cos_r2, sin_r2 = ops.rope_cos_sin(pos2, inv, scaling, dtype)
    assert torch.equal(emb2, emb_r) and torch.equal(nrm2, nrm_r)
    assert torch.equal(cos2, cos_r2) and torch.equal(sin2, sin_r2)


def test_pass_head_rejects_what_it_cannot_do(ops):
    table = torch.zeros((10, 256), dtype=torch.float16, device=DEV)
    wgt = torch.ones(256, dtype=torch.float16, device=DEV)
    ids = torch.zeros((1, 4), dtype=torch.int64, device=DEV)
    with pytest.raises(ValueError):
        ops.pass_head(table, ids, _inv_freq(), 1.0, wgt, 1e-5)                                     # no positions at all
    with pytest.raises(RuntimeError):
        ops.pass_head(table.cpu(), ids, _inv_freq(), 1.0, wgt, 1e-5, position_ids=ids)              # no CPU path
    assert not ops.pass_head_supported(torch.zeros((1, 200), dtype=torch.int64, device=DEV), table, wgt)   # prefill-sized


def test_generation_is_the_same_with_and_without_the_head_launch(ops, monkeypatch):
    """tree / chain / vanilla loops with ops.PASS_HEAD on (default) and off: identical token ids, counts and caches."""
    import cases
    from longspec_amd.llama_glide import LlamaGlide
    run = [r for r in cases.generate_runs() if r["name"] == "mixed"][0]

    def go(flag):
        monkeypatch.setattr(ops, "PASS_HEAD", flag)
        m = LlamaGlide(run["cfg"], device=DEV)
        m.load_state_dict({**run["target_sd"], **{"glide." + k: v for k, v in run["draft_sd"].items()}}, strict=True)
        pl = torch.tensor([run["prompt_len"]], device=DEV)
        t = m.tree_spec_generate(run["prompt"].to(DEV), pl, tree_shape=run["tree_shape"], max_gen_len=run["max_gen_len"])
        kc = m.model.layers[-1].self_attn.K_Cache.clone()
        c = m.spec_generate(run["prompt"].to(DEV), pl, gamma=4, max_gen_len=run["max_gen_len"])
        v = m.vanilla_generate(run["prompt"].to(DEV), pl, max_gen_len=run["max_gen_len"])
        return t, kc, c, v

    (t1, k1, c1, v1), (t0, k0, c0, v0) = go(True), go(False)
    assert torch.equal(t1[0], t0[0]) and t1[1:3] == t0[1:3] and torch.equal(k1, k0)
    assert torch.equal(c1[0], c0[0]) and c1[1:3] == c0[1:3]
    assert torch.equal(v1[0], v0[0])
    assert torch.equal(t1[0].cpu(), run["tree_out"])
