"""Drop-in boundary at the Python level (SURVEY 8(b), 8(c) last row): the reference's harness runs unchanged.

* the launcher puts the MI355X modules under the names the reference scripts import (``from llama_glide import
  LlamaGlide``, ``inference_long-bench.py:1``) -- CPU test, no compute;
* checkpoint resolution the way ``from_pretrained`` does it: shard index files, safetensors vs .bin, hub ids;
* ``-m gpu``: the call sequence of ``inference_long-bench.py:95-112,232-260`` on a toy checkpoint DIRECTORY --
  ``AutoConfig.from_pretrained`` -> ``LlamaGlide(config, target, draft)`` (three positional arguments, no device) ->
  ``.cuda()`` inputs -> warm-up call + timed calls -> the harness's tau and tokens/s formulas -- token-exact against
  the reference's golden run of the same toy model."""
import json
import os
import subprocess
import sys

import pytest
import torch

import cases
import toy
from conftest import ROOT


def _write_checkpoints(tmp_path, run, sharded=False):
    from safetensors.torch import save_file
    cfg = run["cfg"]
    tdir, ddir = tmp_path / "target", tmp_path / "draft"
    os.makedirs(tdir), os.makedirs(ddir)
    hf = {k: v for k, v in vars(cfg).items() if isinstance(v, (int, float, str, bool, list, dict, type(None)))}
    hf.update(model_type="llama", architectures=["LlamaForCausalLM"], torch_dtype="float16")
    json.dump(hf, open(tdir / "config.json", "w"))
    json.dump(hf, open(ddir / "config.json", "w"))
    tgt = {k: v.contiguous() for k, v in run["target_sd"].items()}
    if sharded:
        names = sorted(tgt)
        half = len(names) // 2
        shards = {"model-00001-of-00002.safetensors": names[:half], "model-00002-of-00002.safetensors": names[half:]}
        for fn, ks in shards.items():
            save_file({k: tgt[k] for k in ks}, str(tdir / fn))
        json.dump({"metadata": {}, "weight_map": {k: fn for fn, ks in shards.items() for k in ks}},
                  open(tdir / "model.safetensors.index.json", "w"))
        # a stale consolidated .bin next to the shards must not be picked up
        torch.save({"model.embed_tokens.weight": torch.zeros(1)}, str(tdir / "pytorch_model.bin"))
    else:
        save_file(tgt, str(tdir / "model.safetensors"))
    save_file({k: v.contiguous() for k, v in run["draft_sd"].items()}, str(ddir / "model.safetensors"))
    return str(tdir), str(ddir)


def test_launcher_aliases_the_reference_module_names(tmp_path):
    script = tmp_path / "inference_fake.py"
    script.write_text(
        "from llama_glide import LlamaGlide\n"
        "from qwen2_glide import Qwen2Glide\n"
        "import sys\n"
        "print('MOD', LlamaGlide.__module__, Qwen2Glide.__module__, sys.argv[1:])\n")
    # a decoy sibling module, as in the reference tree: the script's own directory is first on sys.path
    (tmp_path / "llama_glide.py").write_text("raise ImportError('the reference module was imported')\n")
    out = subprocess.run([sys.executable, "-m", "longspec_amd.harness", str(script), "--model_name", "llama8b"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "MOD longspec_amd.llama_glide longspec_amd.qwen2_glide ['--model_name', 'llama8b']" in out.stdout


def test_shims_reexport_the_public_classes():
    sys.path.insert(0, os.path.join(ROOT, "shims"))
    try:
        for name in ("llama_glide", "qwen2_glide"):
            sys.modules.pop(name, None)
        import llama_glide
        import qwen2_glide
        from longspec_amd.llama_glide import LlamaGlide
        from longspec_amd.qwen2_glide import Qwen2Glide
        assert llama_glide.LlamaGlide is LlamaGlide and qwen2_glide.Qwen2Glide is Qwen2Glide
    finally:
        sys.path.remove(os.path.join(ROOT, "shims"))
        for name in ("llama_glide", "qwen2_glide"):
            sys.modules.pop(name, None)


def test_checkpoint_files_follow_the_shard_index(tmp_path):
    from longspec_amd import checkpoint
    import oracle_ops
    from longspec_amd.llama_glide import LlamaGlide
    run = [r for r in cases.generate_runs() if r["name"] == "mixed"][0]
    tdir, ddir = _write_checkpoints(tmp_path, run, sharded=True)
    files = checkpoint.checkpoint_files(tdir)
    assert [os.path.basename(f) for f in files] == ["model-00001-of-00002.safetensors", "model-00002-of-00002.safetensors"]
    m = LlamaGlide(checkpoint.load_config(tdir), tdir, ddir, ops=oracle_ops)
    sd = m.state_dict()
    for k, v in run["target_sd"].items():
        assert torch.equal(sd[k], v)
    os.remove(os.path.join(tdir, "model-00002-of-00002.safetensors"))
    with pytest.raises(FileNotFoundError, match="shard files that are missing"):
        checkpoint.checkpoint_files(tdir)


def test_hub_ids_resolve_through_the_hf_cache_or_fail_loudly(tmp_path, monkeypatch):
    from longspec_amd import checkpoint
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    with pytest.raises(FileNotFoundError, match="neither a local checkpoint path nor a hub repository"):
        checkpoint.resolve_path("sail/longspec-Llama-3-8B-Instruct-262k")
    # a populated cache entry is found without network access
    import huggingface_hub
    snap = tmp_path / "hf" / "hub" / "models--sail--toy-draft" / "snapshots" / "abc123"
    os.makedirs(snap)
    (snap / "config.json").write_text("{}")
    refs = tmp_path / "hf" / "hub" / "models--sail--toy-draft" / "refs"
    os.makedirs(refs)
    (refs / "main").write_text("abc123")
    monkeypatch.setattr(huggingface_hub.constants, "HF_HUB_CACHE", str(tmp_path / "hf" / "hub"), raising=False)
    got = huggingface_hub.snapshot_download("sail/toy-draft", local_files_only=True, cache_dir=str(tmp_path / "hf" / "hub"))
    assert os.path.samefile(got, snap)


@pytest.mark.gpu
def test_harness_call_sequence_on_a_toy_checkpoint(tmp_path):
    """inference_long-bench.py:95-112 (config, model construction) and :232-260 (warm-up, timed loop, formulas)."""
    from transformers import AutoConfig
    from longspec_amd.harness import install_aliases
    install_aliases()
    from llama_glide import LlamaGlide                      # the harness's import line (:1)
    run = [r for r in cases.generate_runs() if r["name"] == "mixed"][0]
    target_model_name, draft_model_name = _write_checkpoints(tmp_path, run)
    config = AutoConfig.from_pretrained(target_model_name)           # :104  (a transformers LlamaConfig object)
    config.pad_token_id = run["cfg"].pad_token_id                    # :109-111 sets ids on the config object
    config.eos_token_id = run["cfg"].eos_token_id
    llama_glide = LlamaGlide(config, target_model_name, draft_model_name)      # :112 -- three positional arguments
    assert next(llama_glide.parameters()).is_cuda and next(llama_glide.parameters()).dtype == torch.float16
    a_cuda = run["prompt"].cuda()                                    # :118-119
    len_cuda = torch.tensor([run["prompt_len"]]).cuda()
    meta_prompts = [{"input_ids": a_cuda, "length": len_cuda}] * 2
    counts = nums = 0
    glide_time = .0
    with torch.inference_mode():
        # warm up (:233-240)
        output_ids, count, num, elapsed_time, spec_mask = llama_glide.tree_spec_generate(
            meta_prompts[0]["input_ids"], prompt_length=meta_prompts[0]["length"], max_gen_len=run["max_gen_len"],
            tree_shape=run["tree_shape"], temperature=0.0)
        # real run (:242-256)
        for i in range(2):
            meta_prompt = meta_prompts[i]
            output_ids, count, num, elapsed_time, spec_mask = llama_glide.tree_spec_generate(
                meta_prompt["input_ids"], prompt_length=meta_prompt["length"], max_gen_len=run["max_gen_len"],
                tree_shape=run["tree_shape"], temperature=0.0)
            assert torch.equal(output_ids.cpu(), run["tree_out"])
            assert (int(count), int(num)) == (run["tree_count"], run["tree_num"])
            assert elapsed_time > 0
            glide_time += elapsed_time
            counts += count
            nums += num
    tau = (counts + nums) / nums                                     # :259
    tok_s = (counts + nums) / glide_time                             # :260
    assert float(tau) == pytest.approx((run["tree_count"] + run["tree_num"]) / run["tree_num"])
    assert float(tok_s) > 0
    # the other --method branches of the harness on the same object (:132-230)
    with torch.inference_mode():
        out_v, num_v, t_v = llama_glide.vanilla_generate(a_cuda, prompt_length=len_cuda, max_gen_len=run["max_gen_len"])
        assert torch.equal(out_v.cpu(), run["vanilla_out"]) and num_v == run["vanilla_num"]
        out_s, c_s, n_s, t_s, _ = llama_glide.spec_generate(a_cuda, prompt_length=len_cuda, max_gen_len=run["max_gen_len"],
                                                            gamma=4, temperature=0.0)
        assert (int(c_s), int(n_s)) == (run["chain_count"], run["chain_num"])
