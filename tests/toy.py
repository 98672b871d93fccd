"""Deterministic toy models and synthetic inputs shared by the golden-vector
generator (``tests/golden/make_golden.py``) and the tests.

Weights come from a seeded CPU ``torch.Generator`` in a fixed (sorted-name)
order so the generator (which loads them into the *reference* model) and the
tests (which load them into ``longspec_amd``) see identical tensors without
committing megabytes of weights.  Every fixture stores a checksum of what was
generated so that RNG drift is detected instead of silently mis-compared.
"""
from __future__ import annotations

import hashlib
from types import SimpleNamespace
from typing import Dict, List

import numpy as np
import torch

# head_dim must be 128: the reference hard-codes the target's softmax scale to
# 1/sqrt(128) (longspec/test/llama.py:95).
TOY_CFG = dict(
    hidden_size=256,
    intermediate_size=384,
    num_hidden_layers=2,
    num_attention_heads=2,
    num_key_value_heads=1,
    vocab_size=512,
    max_position_embeddings=4096,
    rms_norm_eps=1e-5,
    rope_theta=10000.0,
    pad_token_id=511,      # outside the sampled vocab
    eos_token_id=510,
    bos_token_id=1,
)

TOY_CFG_GQA = dict(TOY_CFG, hidden_size=512, num_attention_heads=4, num_key_value_heads=2)


def toy_config(**overrides) -> SimpleNamespace:
    d = dict(TOY_CFG)
    d.update(overrides)
    d.setdefault("head_dim", d["hidden_size"] // d["num_attention_heads"])
    d.setdefault("attention_bias", False)
    d.setdefault("mlp_bias", False)
    d.setdefault("hidden_act", "silu")
    d.setdefault("rope_scaling", None)
    return SimpleNamespace(**d)


def target_param_shapes(cfg) -> Dict[str, tuple]:
    Hd, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    H, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    D = Hd // H
    shapes = {"model.embed_tokens.weight": (V, Hd), "model.norm.weight": (Hd,), "lm_head.weight": (V, Hd)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        shapes.update({
            p + "self_attn.q_proj.weight": (H * D, Hd),
            p + "self_attn.k_proj.weight": (Hkv * D, Hd),
            p + "self_attn.v_proj.weight": (Hkv * D, Hd),
            p + "self_attn.o_proj.weight": (Hd, H * D),
            p + "mlp.gate_proj.weight": (I, Hd),
            p + "mlp.up_proj.weight": (I, Hd),
            p + "mlp.down_proj.weight": (Hd, I),
            p + "input_layernorm.weight": (Hd,),
            p + "post_attention_layernorm.weight": (Hd,),
        })
        if getattr(cfg, "attention_bias", False):          # Qwen2 target: q/k/v bias (qwen2.py:277-279)
            shapes.update({p + "self_attn.q_proj.bias": (H * D,), p + "self_attn.k_proj.bias": (Hkv * D,),
                           p + "self_attn.v_proj.bias": (Hkv * D,)})
    return shapes


def draft_param_shapes(cfg) -> Dict[str, tuple]:
    """The 20 tensors of the draft checkpoint: q/k/v have
    bias=True even for Llama (longspec/test/llama_glide.py:49-52)."""
    Hd, I = cfg.hidden_size, cfg.intermediate_size
    H, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    D = Hd // H
    shapes = {}
    for a in ("self_attn", "cross_attn"):
        shapes.update({
            f"{a}.q_proj.weight": (H * D, Hd), f"{a}.q_proj.bias": (H * D,),
            f"{a}.k_proj.weight": (Hkv * D, Hd), f"{a}.k_proj.bias": (Hkv * D,),
            f"{a}.v_proj.weight": (Hkv * D, Hd), f"{a}.v_proj.bias": (Hkv * D,),
            f"{a}.o_proj.weight": (Hd, H * D),
        })
    shapes.update({
        "mlp.gate_proj.weight": (I, Hd), "mlp.up_proj.weight": (I, Hd), "mlp.down_proj.weight": (Hd, I),
        "input_layernorm.weight": (Hd,), "post_self_attention_layernorm.weight": (Hd,),
        "post_cross_attention_layernorm.weight": (Hd,),
    })
    return shapes


def _fill(shapes: Dict[str, tuple], gen: torch.Generator, std: float, agreement: float) -> Dict[str, torch.Tensor]:
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith("layernorm.weight") or name.endswith("norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=gen)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shp, generator=gen)
        else:
            t = std * torch.randn(shp, generator=gen)
        # forced / mixed agreement: scale every o_proj and
        # down_proj so logits depend (mostly) on the current token only.
        if agreement != 1.0 and (name.endswith("o_proj.weight") or name.endswith("down_proj.weight")):
            t = t * agreement
        out[name] = t.to(torch.float16)
    return out


def make_weights(cfg, seed: int, agreement: float = 1.0, std: float = 0.05):
    """Returns (target_state_dict, draft_state_dict) in fp16.
    agreement = 1.0: plain random model (tau ~ 1); 0.0: forced agreement (every
    o_proj/down_proj zero => full-depth acceptance); small eps: mixed acceptance."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    tgt = _fill(target_param_shapes(cfg), g, std, agreement)
    drf = _fill(draft_param_shapes(cfg), g, std, agreement)
    return tgt, drf


def make_prompt(cfg, length: int, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randint(2, cfg.vocab_size - 4, (1, length), generator=g, dtype=torch.int64)


def checksum(*tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        if torch.is_tensor(t):
            t = t.detach().cpu().contiguous()
            t = t.view(torch.int16).numpy() if t.dtype in (torch.float16, torch.bfloat16) else t.numpy()
        h.update(np.ascontiguousarray(t).tobytes())
    return h.hexdigest()[:16]


def state_checksum(sd: Dict[str, torch.Tensor]) -> str:
    return checksum(*[sd[k] for k in sorted(sd)])


# --------------------------------------------------------------------------- #
# synthetic trees / kernel-level inputs
# --------------------------------------------------------------------------- #
def level_sizes(tree_shape: List[int]) -> List[int]:
    acc = [1]
    for c in tree_shape:
        acc.append(acc[-1] + c)
    return acc


def random_beam_tree(tree_shape: List[int], seed: int) -> np.ndarray:
    """Parents of a random beam tree in level order: node 0 = root, parent of each
    level-l node uniform over level l-1.  Returns parents [F] (parents[0] = 0)."""
    rng = np.random.RandomState(seed)
    acc = level_sizes(tree_shape)
    parents = np.zeros(acc[-1], dtype=np.int64)
    for lvl in range(1, len(acc)):
        lo, hi = acc[lvl - 1], acc[lvl]
        plo = 0 if lvl == 1 else acc[lvl - 2]
        phi = acc[lvl - 1]
        parents[lo:hi] = np.sort(rng.randint(plo, phi, size=hi - lo))
    return parents


def tree_mask_from_parents(parents: np.ndarray) -> np.ndarray:
    """tree_mask[r] = ancestors of r incl. r and the root (column 0), int64 [F,F]
    -- the same matrix tree_spec_generate builds (llama_glide.py:1022,1069-1071)."""
    Fn = len(parents)
    m = np.zeros((Fn, Fn), dtype=np.int64)
    m[:, 0] = 1
    for r in range(1, Fn):
        m[r] = m[parents[r]]
        m[r, r] = 1
    return m


def verify_mask(tree_mask: np.ndarray, a: int, gamma: int) -> np.ndarray:
    """The 74x74 verification mask ``new_tree_mask`` (llama_glide.py:1082-1085):
    tril(ones) with the (F-1)x(F-1) non-root tree block at [a:a+F-1]^2."""
    Fn = tree_mask.shape[0]
    R = Fn - 1 + gamma + 1
    m = np.ones((R, R), dtype=np.int64)
    m[a:a + Fn - 1, a:a + Fn - 1] = tree_mask[1:, 1:]
    return np.tril(m)


def randn_f16(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.float16)


def verify_inputs(H, Hkv, L, seed, a=3, max_len=80, tree_shape=(4, 16, 16, 16, 16)):
    """Seeded inputs of one hybrid verification-attention call:
    q/k_new/v_new ~ N(0,1) fp16 [1,R,*,128], prefix K/V ~ N(0,1) in caches of
    L+max_len rows, the 74x74 verify mask of a seeded beam tree with ``a``
    accepted tokens in front."""
    parents = random_beam_tree(list(tree_shape), seed)
    vm = verify_mask(tree_mask_from_parents(parents), a=a, gamma=len(tree_shape))
    R = vm.shape[0]
    q = randn_f16((1, R, H, 128), seed * 7 + 0)
    k = randn_f16((1, R, Hkv, 128), seed * 7 + 1)
    v = randn_f16((1, R, Hkv, 128), seed * 7 + 2)
    kc = torch.zeros(1, L + max_len, Hkv, 128, dtype=torch.float16)
    vc = torch.zeros(1, L + max_len, Hkv, 128, dtype=torch.float16)
    kc[:, :L] = randn_f16((1, L, Hkv, 128), seed * 7 + 3)
    vc[:, :L] = randn_f16((1, L, Hkv, 128), seed * 7 + 4)
    return q, k, v, kc, vc, torch.from_numpy(vm).unsqueeze(0)
