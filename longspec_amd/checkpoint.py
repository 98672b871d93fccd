"""Checkpoint loading in the reference's formats, without the HF ``from_pretrained`` coupling.

* target: a HF Llama directory (``config.json`` + ``*.safetensors`` / ``pytorch_model*.bin``,
  optionally sharded with an ``*.index.json``), as ``LlamaForCausalLM.from_pretrained`` reads
  it (``longspec/test/llama_glide.py:474``);
* draft: the directory ``LlamaGlideDecoderLayer.from_pretrained`` reads (``:480``): 20 tensors
  ``{self_attn,cross_attn}.{q,k,v}_proj.{weight,bias}``, ``{self_attn,cross_attn}.o_proj.weight``,
  ``mlp.{gate,up,down}_proj.weight``, ``{input,post_self_attention,post_cross_attention}_layernorm.weight``
  (SURVEY section 5); also accepts the trainer's ``draft_model_weights.pth`` state dict
  (``longspec/train/trainer_base_ds_mul_fs_tp.py:49-113``).
"""
from __future__ import annotations

import glob
import json
import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

DRAFT_TENSORS = (
    [f"{a}.{p}_proj.{w}" for a in ("self_attn", "cross_attn") for p in ("q", "k", "v") for w in ("weight", "bias")]
    + [f"{a}.o_proj.weight" for a in ("self_attn", "cross_attn")]
    + [f"mlp.{p}_proj.weight" for p in ("gate", "up", "down")]
    + [f"{n}.weight" for n in ("input_layernorm", "post_self_attention_layernorm", "post_cross_attention_layernorm")]
)


def load_config(path: str) -> SimpleNamespace:
    path = resolve_path(path, need_tensors=False)
    with open(os.path.join(path, "config.json")) as f:
        d = json.load(f)
    d.setdefault("head_dim", d["hidden_size"] // d["num_attention_heads"])
    d.setdefault("attention_bias", False)
    d.setdefault("mlp_bias", False)
    d.setdefault("rope_theta", 10000.0)
    d.setdefault("rope_scaling", None)
    d.setdefault("num_key_value_heads", d["num_attention_heads"])
    return SimpleNamespace(**d)


def _snapshot_complete(path: str) -> bool:
    """Tensors present, and ALL of them: an index that names absent shards is incomplete, and so is a directory of
    ``model-0000i-of-0000N`` shards (no index: an interrupted download) that does not hold all N."""
    import re
    try:
        files = checkpoint_files(path)
    except FileNotFoundError:                    # an index that names absent shards
        return False
    if not files:
        return False
    shards = {}
    for f in files:
        m = re.search(r"-(\d+)-of-(\d+)\.(safetensors|bin)$", os.path.basename(f))
        if m:
            shards.setdefault(int(m.group(2)), set()).add(int(m.group(1)))
    return all(len(have) == total for total, have in shards.items())


def resolve_path(path_or_id: str, need_tensors: bool = True, revision: Optional[str] = None) -> str:
    """A local checkpoint directory / file as is; anything else is taken as a Hugging Face hub id
    (``lmsys/vicuna-7b-v1.5-16k``, ``sail/longspec-vicuna-7b-v1.5-16k`` ... -- what
    ``inference_long-bench.py:41-62`` passes to ``from_pretrained``) and resolved through the local HF cache first,
    then -- if the machine has network access -- downloaded with ``snapshot_download``.

    The cache wins: a COMPLETE cached snapshot (or, for ``need_tensors=False``, a cached ``config.json``) is returned without
    asking the hub, so a repository updated since is not refreshed.  To force a refresh pass ``revision`` (a commit
    hash / tag / branch: the cache is keyed by it) or clear the cache entry; with ``HF_HUB_OFFLINE=1`` huggingface_hub itself
    refuses the second, networked pass."""
    if os.path.exists(path_or_id):
        return path_or_id
    try:
        from huggingface_hub import snapshot_download
    except ImportError as e:                     # pragma: no cover
        raise FileNotFoundError(f"{path_or_id!r} is not a local path and huggingface_hub is not installed") from e
    patterns = ["*.json", "*.safetensors", "*.bin", "*.pth", "*.pt", "*.model"]
    rev = {"revision": revision} if revision is not None else {}
    cached = None
    try:
        cached = snapshot_download(path_or_id, local_files_only=True, allow_patterns=patterns, **rev)
        # A snapshot can be PARTIAL: `AutoConfig.from_pretrained(hub_id)` (inference_long-bench.py:104) caches config.json alone,
        # and the local-only pass cannot tell (no tree listing is cached).  Only a snapshot that holds its tensors -- every shard
        # its index names -- is complete; anything else goes on to the networked pass, which fetches what is missing.
        if _snapshot_complete(cached):
            return cached
        if not need_tensors and os.path.exists(os.path.join(cached, "config.json")):
            return cached                        # load_config: the cached config.json is all it reads -- no multi-GB download for it
    except Exception:
        pass
    try:
        return snapshot_download(path_or_id, allow_patterns=patterns, **rev)
    except Exception as e:
        if cached is not None and os.path.exists(os.path.join(cached, "config.json")) and not need_tensors:
            return cached                        # config-only consumers (load_config) can live with the partial snapshot
        raise FileNotFoundError(
            f"{path_or_id!r} is neither a local checkpoint path nor a hub repository reachable from this machine "
            f"({'only a partial snapshot (no tensors) is' if cached else 'not'} in the HF cache, download failed: {type(e).__name__}).  Download it once "
            f"(`huggingface-cli download {path_or_id}`) or pass a local directory.") from e


def checkpoint_files(path: str):
    """Tensor files of a checkpoint directory, the way ``from_pretrained`` picks them: a shard index
    (``model.safetensors.index.json`` / ``pytorch_model.bin.index.json``) names the shard files; without one,
    ``*.safetensors`` is preferred over ``pytorch_model*.bin`` over raw ``*.pth`` / ``*.pt`` state dicts."""
    if os.path.isfile(path):
        return [path]
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        ip = os.path.join(path, index)
        if os.path.exists(ip):
            with open(ip) as f:
                names = sorted(set(json.load(f)["weight_map"].values()))
            files = [os.path.join(path, n) for n in names]
            absent = [f for f in files if not os.path.exists(f)]
            if absent:
                raise FileNotFoundError(f"{ip} names shard files that are missing: {absent[:3]}")
            return files
    for pat in ("*.safetensors", "pytorch_model*.bin", "*.pth", "*.pt"):
        files = sorted(glob.glob(os.path.join(path, pat)))
        if files:
            return files
    return []


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """All tensors of a checkpoint directory, single file or hub id."""
    path = resolve_path(path)
    files = checkpoint_files(path)
    if not files:
        raise FileNotFoundError(f"no checkpoint tensors under {path}")
    sd: Dict[str, torch.Tensor] = {}
    for fp in files:
        if fp.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd.update(load_file(fp))
        else:
            sd.update(torch.load(fp, map_location="cpu", weights_only=True))
    return sd


def load_target_checkpoint(model, path: str) -> None:
    sd = read_state_dict(path)
    if "lm_head.weight" not in sd and "model.embed_tokens.weight" in sd:      # tied embeddings
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    own = {k for k in model.state_dict() if not k.startswith("glide.")}
    missing = sorted(own - set(sd))
    if missing:
        raise KeyError(f"target checkpoint {path} lacks {missing[:5]}{'...' if len(missing) > 5 else ''}")
    model.load_state_dict({k: sd[k] for k in own}, strict=False)


def load_draft_checkpoint(glide, path: str) -> None:
    sd = read_state_dict(path)
    for prefix in ("draft_model.", "glide.", "model."):
        if any(k.startswith(prefix) for k in sd) and not all(k in sd for k in DRAFT_TENSORS):
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    missing = [k for k in DRAFT_TENSORS if k not in sd]
    if missing:
        raise KeyError(f"draft checkpoint {path} lacks {missing}")
    glide.load_state_dict({k: sd[k] for k in DRAFT_TENSORS}, strict=True)


def save_draft_checkpoint(glide, path: str, config=None) -> None:
    """Write the draft layer in the reference's directory format (safetensors + config.json)."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    sd = {k: v.detach().cpu().contiguous() for k, v in glide.state_dict().items() if k in DRAFT_TENSORS}
    save_file(sd, os.path.join(path, "model.safetensors"))
    if config is not None:
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump({k: v for k, v in vars(config).items() if isinstance(v, (int, float, str, bool, list, dict, type(None)))}, f)
