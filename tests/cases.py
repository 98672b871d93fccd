"""Rebuild the inputs of every golden case (same seeds as
``tests/golden/make_golden.py``) and pair them with the reference's outputs.
Used by the CPU oracle tests and by the ``-m gpu`` parity tests alike."""
from __future__ import annotations

import numpy as np
import torch

import toy
from conftest import cksum_str, load_golden

TREE = [4, 16, 16, 16, 16]
LEVELS = ((4, 5), (16, 21), (16, 37), (16, 53))


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if dtype is not None else t


def triton_cases():
    g = load_golden("triton_tree_attn")
    for i in range(int(g["n_cases"])):
        t = f"c{i}"
        H, Hkv, M, N, seed = (int(g[f"{t}_{k}"]) for k in ("H", "Hkv", "M", "N", "seed"))
        q = toy.randn_f16((1, H, M, 128), seed * 3 + 0)
        k = toy.randn_f16((1, Hkv, N, 128), seed * 3 + 1)
        v = toy.randn_f16((1, Hkv, N, 128), seed * 3 + 2)
        assert toy.checksum(q, k, v) == cksum_str(g[f"{t}_in_checksum"]), "RNG drift: regenerate goldens"
        yield dict(name=t, q=q, k=k, v=v, mask=_t(g[f"{t}_mask"], torch.int64), o=_t(g[f"{t}_o"]), L=_t(g[f"{t}_L"]))


def tree_part_cases():
    g = load_golden("target_tree_part")
    for i in range(int(g["n_cases"])):
        t = f"c{i}"
        H, Hkv, seed = (int(g[f"{t}_{k}"]) for k in ("H", "Hkv", "seed"))
        q, k, v, kc, vc, tm = toy.verify_inputs(H, Hkv, 64, seed)
        assert toy.checksum(q, k, v) == cksum_str(g[f"{t}_in_checksum"]), "RNG drift: regenerate goldens"
        yield dict(name=t, q=q, k=k, v=v, kc=kc, vc=vc, mask=tm, prefix_lse=_t(g[f"{t}_prefix_lse"]),
                   last_layer=bool(int(g[f"{t}_last_layer"])), current_out=_t(g[f"{t}_current_out"]),
                   weight=_t(g[f"{t}_weight"]), kcache_after=_t(g[f"{t}_kcache_after"]),
                   vcache_after=_t(g[f"{t}_vcache_after"]))


def verify_cases():
    g = load_golden("verify_attention")
    for i in range(int(g["n_cases"])):
        t = f"c{i}"
        H, Hkv, L, seed = (int(g[f"{t}_{k}"]) for k in ("H", "Hkv", "L", "seed"))
        q, k, v, kc, vc, tm = toy.verify_inputs(H, Hkv, L, seed)
        assert toy.checksum(q, k, v, kc, vc) == cksum_str(g[f"{t}_in_checksum"]), "RNG drift: regenerate goldens"
        d = dict(name=t, H=H, Hkv=Hkv, L=L, q=q, k=k, v=v, kc=kc, vc=vc, mask=tm,
                 cache_lens=torch.tensor([L], dtype=torch.int32),
                 hybrid={False: _t(g[f"{t}_hybrid_last0"]), True: _t(g[f"{t}_hybrid_last1"])})
        d["dense"] = _t(g[f"{t}_dense"]) if int(g["have_dense"]) else None
        yield d


def decoding_torch_cases():
    """G-h: outputs of the reference's own ``LlamaAttention.decoding_torch`` (longspec/test/llama.py:161-197) on seeded
    inputs (tests/golden/make_golden.py::gen_decoding_torch)."""
    g = load_golden("decoding_torch")
    for i in range(int(g["n_cases"])):
        t = f"c{i}"
        H, Hkv, L, a, seed = (int(g[f"{t}_{k}"]) for k in ("H", "Hkv", "L", "a", "seed"))
        q = toy.randn_f16((1, a, H, 128), seed * 13 + 0)
        k = toy.randn_f16((1, a, Hkv, 128), seed * 13 + 1)
        v = toy.randn_f16((1, a, Hkv, 128), seed * 13 + 2)
        kc = torch.zeros(1, L + 16, Hkv, 128, dtype=torch.float16)
        vc = torch.zeros(1, L + 16, Hkv, 128, dtype=torch.float16)
        kc[:, :L] = toy.randn_f16((1, L, Hkv, 128), seed * 13 + 3)
        vc[:, :L] = toy.randn_f16((1, L, Hkv, 128), seed * 13 + 4)
        assert toy.checksum(q, k, v, kc, vc) == cksum_str(g[f"{t}_in_checksum"]), "RNG drift: regenerate goldens"
        yield dict(name=t, H=H, Hkv=Hkv, L=L, a=a, q=q, k=k, v=v, kc=kc, vc=vc, out=_t(g[f"{t}_out"]),
                   k_rows=_t(g[f"{t}_kcache_rows"]), v_rows=_t(g[f"{t}_vcache_rows"]))


def draft_cases():
    """Each case: a draft self-attn KV cache with p valid rows, then step 0
    (a = 3 rows appended at p-2) and the four tree steps, chained on the same cache."""
    g = load_golden("draft_attention")
    for i in range(int(g["n_cases"])):
        t = f"c{i}"
        H, Hkv, p, seed = (int(g[f"{t}_{k}"]) for k in ("H", "Hkv", "p", "seed"))
        kc = torch.zeros(1, p + 200, Hkv, 128, dtype=torch.float16)
        vc = torch.zeros(1, p + 200, Hkv, 128, dtype=torch.float16)
        kc[:, :p] = toy.randn_f16((1, p, Hkv, 128), seed * 11 + 0)
        vc[:, :p] = toy.randn_f16((1, p, Hkv, 128), seed * 11 + 1)
        assert toy.checksum(kc, vc) == cksum_str(g[f"{t}_in_checksum"]), "RNG drift: regenerate goldens"
        steps = [dict(kind="step0", q=_t(g[f"{t}_s0_q"]), k=_t(g[f"{t}_s0_k"]), v=_t(g[f"{t}_s0_v"]),
                      cache_lens=_t(g[f"{t}_s0_cache_lens"]), out=_t(g[f"{t}_s0_out"]))]
        for lvl in range(4):
            steps.append(dict(kind="tree", q=_t(g[f"{t}_t{lvl}_q"]), k=_t(g[f"{t}_t{lvl}_k"]), v=_t(g[f"{t}_t{lvl}_v"]),
                              mask=_t(g[f"{t}_t{lvl}_mask"], torch.int64),
                              cache_lens=torch.tensor([p], dtype=torch.int32), out=_t(g[f"{t}_t{lvl}_out"])))
        yield dict(name=t, H=H, Hkv=Hkv, p=p, kc=kc, vc=vc, steps=steps)


def tree_verification_cases():
    g = load_golden("tree_verification")
    for i in range(int(g["n_cases"])):
        t = f"c{i}"
        yield dict(name=t, spec=_t(g[f"{t}_spec"])[None], pred=_t(g[f"{t}_pred"])[None],
                   mask=_t(g[f"{t}_mask"], torch.int64)[None], non_leaf_len=int(g[f"{t}_non_leaf_len"]),
                   cache_len=int(g[f"{t}_cache_len"]), kc=_t(g[f"{t}_kc"]), vc=_t(g[f"{t}_vc"]),
                   acc_ids=_t(g[f"{t}_acc_ids"]), acc_num=_t(g[f"{t}_acc_num"]),
                   double_input=_t(g[f"{t}_double_input"]), kc_after=_t(g[f"{t}_kc_after"]),
                   vc_after=_t(g[f"{t}_vc_after"]))


def norm_cases():
    g = load_golden("norm_rope")
    for i in range(int(g["n_norm"])):
        yield dict(name=f"norm{i}", x=_t(g[f"norm{i}_x"]), w=_t(g[f"norm{i}_w"]), eps=float(g[f"norm{i}_eps"]),
                   y=_t(g[f"norm{i}_y"]))


def rope_cases():
    g = load_golden("norm_rope")
    for i in range(int(g["n_rope"])):
        p = f"rope{i}_"
        yield dict(name=f"rope{i}", inv_freq=_t(g[p + "inv_freq"]), scaling=float(g[p + "scaling"]),
                   pos=_t(g[p + "pos"]), cos=_t(g[p + "cos"]), sin=_t(g[p + "sin"]), q=_t(g[p + "q"]), k=_t(g[p + "k"]),
                   q_out=_t(g[p + "q_out"]), k_out=_t(g[p + "k_out"]), theta=float(g[p + "theta"]),
                   factor=float(g[p + "factor"]))


def generate_runs(family="llama"):
    """family: "llama", "qwen2", or "qwen2_bf16" (the Qwen2 twin in bfloat16, as inference_qwq.py runs QwQ); with the
    suffix "_long" the runs of the regime every BASELINE configuration is in (prompt >= 700 tokens: the draft's 512-row
    window truncates from round 1; >= 64 rounds; three seeds per weight kind -- make_golden.py::gen_generate_long)."""
    dtype = torch.bfloat16 if family.startswith("qwen2_bf16") else torch.float16
    g = load_golden({"llama": "generate", "qwen2": "generate_qwen2", "qwen2_bf16": "generate_qwen2_bf16",
                     "llama_long": "generate_long", "qwen2_long": "generate_long_qwen2",
                     "qwen2_bf16_long": "generate_long_qwen2_bf16"}[family])
    family = "llama" if family.startswith("llama") else "qwen2"
    for name in [str(x) for x in g["runs"]]:
        over = {str(k): int(v) for k, v in zip(g[f"{name}_cfg_keys"], g[f"{name}_cfg_vals"])}
        cfg = toy.toy_config(**over)
        wseed = int(g[f"{name}_wseed"])
        agree = float(g[f"{name}_agreement"])
        tgt, drf = toy.make_weights(cfg, wseed, agreement=agree)
        assert toy.state_checksum(tgt) + toy.state_checksum(drf) == cksum_str(g[f"{name}_weights_checksum"]), \
            "RNG drift: regenerate goldens"
        d = dict(name=name, cfg=cfg, target_sd=tgt, draft_sd=drf, prompt=_t(g[f"{name}_prompt"]),
                 prompt_len=int(g[f"{name}_prompt_len"]), max_gen_len=int(g[f"{name}_max_gen_len"]),
                 tree_shape=[int(x) for x in g[f"{name}_tree_shape"]], family=family, dtype=dtype,
                 eos_id=int(g[f"{name}_eos_id"]) if f"{name}_eos_id" in g else 151645)
        for k in ("vanilla_out", "tree_out", "chain_out", "tr_tree_mask", "tr_all_spec", "tr_llm_pred", "tr_acc_ids",
                  "tr_acc_num", "tr_cache_lens"):
            d[k] = _t(g[f"{name}_{k}"])
        for k in ("vanilla_num", "tree_count", "tree_num", "chain_count", "chain_num"):
            d[k] = int(g[f"{name}_{k}"])
        for k in ("vanilla_top2_ids", "vanilla_top2_logits"):         # long runs: the reference's own decision margins
            if f"{name}_{k}" in g:
                d[k] = _t(g[f"{name}_{k}"])
        yield d


def baseline_runs():
    """The harness's comparison baselines (--method magicdec / vanilla_torch), goldens from the reference."""
    g = load_golden("baselines")
    for name in [str(x) for x in g["runs"]]:
        cfg = toy.toy_config()
        wseed = int(g[f"{name}_wseed"])
        tgt, drf = toy.make_weights(cfg, wseed, agreement=float(g[f"{name}_agreement"]))
        assert toy.state_checksum(tgt) + toy.state_checksum(drf) == cksum_str(g[f"{name}_weights_checksum"]), \
            "RNG drift: regenerate goldens"
        d = dict(name=name, cfg=cfg, target_sd=tgt, draft_sd=drf, prompt=_t(g[f"{name}_prompt"]), family="llama",
                 prompt_len=int(g[f"{name}_prompt_len"]), max_gen_len=int(g[f"{name}_max_gen_len"]), gamma=int(g[f"{name}_gamma"]))
        for k in ("vanilla_out", "vanilla_torch_out", "magicdec_out"):
            d[k] = _t(g[f"{name}_{k}"])
        for k in ("vanilla_torch_num", "magicdec_count", "magicdec_num"):
            d[k] = int(g[f"{name}_{k}"])
        yield d


def stochastic_inputs(ci):
    """Seeded inputs of verify_stochastic unit case ci (same construction as tests/golden/make_golden.py)."""
    shapes = [[4, 16, 16, 16, 16]] * 12 + [[2, 2, 2]] * 4 + [[4, 4]] * 3 + [[1, 1, 1, 1, 1, 1]] * 3 + [[3]] * 2
    shape = shapes[ci]
    parents = toy.random_beam_tree(shape, 9500 + ci)
    mask = toy.tree_mask_from_parents(parents)
    Fn = mask.shape[0]
    V = 160
    g = torch.Generator().manual_seed(9600 + ci)
    spec = torch.randint(2, V, (1, Fn), generator=g)
    logits = (torch.randn(1, Fn, V, generator=g) * 2.0).to(torch.float16)
    spec_logp = (torch.randn(1, Fn, V, generator=g) * 2.0).log_softmax(dim=-1)
    temperature = [0.5, 1.0, 1.3, 0.8][ci % 4]
    return shape, spec, torch.from_numpy(mask)[None].to(torch.int64), logits, spec_logp, temperature


def stochastic_cases():
    g = load_golden("verify_stochastic")
    for ci in range(int(g["n_cases"])):
        shape, spec, mask, logits, spec_logp, T = stochastic_inputs(ci)
        assert toy.checksum(spec, mask, logits, spec_logp) == cksum_str(g[f"c{ci}_in_checksum"]), "RNG drift: regenerate goldens"
        yield dict(name=f"c{ci}", ci=ci, shape=shape, spec=spec, mask=mask, logits=logits, spec_logp=spec_logp, T=T,
                   acc_ids=_t(g[f"c{ci}_acc_ids"]), acc_num=_t(g[f"c{ci}_acc_num"]), after_random=float(g[f"c{ci}_after_random"]))


def stochastic_runs(long=False):
    g = load_golden("verify_stochastic_long" if long else "verify_stochastic")
    for name in [str(x) for x in g["runs"]]:
        over = {str(k): int(v) for k, v in zip(g[f"{name}_cfg_keys"], g[f"{name}_cfg_vals"])}
        cfg = toy.toy_config(**over)
        wseed = int(g[f"{name}_wseed"])
        tgt, drf = toy.make_weights(cfg, wseed, agreement=float(g[f"{name}_agreement"]))
        assert toy.state_checksum(tgt) + toy.state_checksum(drf) == cksum_str(g[f"{name}_weights_checksum"]), \
            "RNG drift: regenerate goldens"
        yield dict(name=name, cfg=cfg, target_sd=tgt, draft_sd=drf, prompt=_t(g[f"{name}_prompt"]), wseed=wseed,
                   prompt_len=int(g[f"{name}_prompt_len"]), max_gen_len=int(g[f"{name}_max_gen_len"]),
                   tree_shape=[int(x) for x in g[f"{name}_tree_shape"]], temperature=float(g[f"{name}_temperature"]),
                   out=_t(g[f"{name}_out"]), count=int(g[f"{name}_count"]), num=int(g[f"{name}_num"]),
                   tr_acc_ids=_t(g[f"{name}_tr_acc_ids"]), tr_acc_num=_t(g[f"{name}_tr_acc_num"]), family="llama")


def chain_stochastic_runs():
    """spec_generate(temperature > 0) of the reference on two toy models (tests/golden/make_golden.py::gen_chain_stochastic)."""
    g = load_golden("chain_stochastic")
    for name in [str(x) for x in g["runs"]]:
        over = {str(k): int(v) for k, v in zip(g[f"{name}_cfg_keys"], g[f"{name}_cfg_vals"])}
        cfg = toy.toy_config(**over)
        wseed = int(g[f"{name}_wseed"])
        tgt, drf = toy.make_weights(cfg, wseed, agreement=float(g[f"{name}_agreement"]))
        assert toy.state_checksum(tgt) + toy.state_checksum(drf) == cksum_str(g[f"{name}_weights_checksum"]), \
            "RNG drift: regenerate goldens"
        yield dict(name=name, cfg=cfg, target_sd=tgt, draft_sd=drf, prompt=_t(g[f"{name}_prompt"]), wseed=wseed,
                   prompt_len=int(g[f"{name}_prompt_len"]), max_gen_len=int(g[f"{name}_max_gen_len"]),
                   temperature=float(g[f"{name}_temperature"]), torch_seed=int(g[f"{name}_torch_seed"]),
                   method=str(g[f"{name}_method"]),
                   out=_t(g[f"{name}_out"]), count=int(g[f"{name}_count"]), num=int(g[f"{name}_num"]), family="llama")
