"""Draft / target checkpoint formats: round trip through the directory layout the
reference's from_pretrained reads, incl. the trainer's prefixed .pth."""
import os

import torch

import oracle_ops
import toy


def test_draft_and_target_checkpoint_roundtrip(tmp_path):
    from safetensors.torch import save_file
    from longspec_amd import checkpoint
    from longspec_amd.llama_glide import LlamaGlide
    cfg = toy.toy_config()
    tgt, drf = toy.make_weights(cfg, 3, agreement=0.05)
    assert sorted(drf) == sorted(checkpoint.DRAFT_TENSORS) and len(drf) == 20
    tdir, ddir, pth = tmp_path / "target", tmp_path / "draft", tmp_path / "draft_model_weights.pth"
    os.makedirs(tdir), os.makedirs(ddir)
    import json
    json.dump({k: v for k, v in vars(cfg).items()}, open(tdir / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in tgt.items()}, str(tdir / "model.safetensors"))
    save_file({k: v.contiguous() for k, v in drf.items()}, str(ddir / "model.safetensors"))
    torch.save({"draft_model." + k: v for k, v in drf.items()}, str(pth))
    cfg2 = checkpoint.load_config(str(tdir))
    assert cfg2.hidden_size == cfg.hidden_size and cfg2.head_dim == 128
    for dpath in (str(ddir), str(pth)):
        m = LlamaGlide(cfg2, str(tdir), dpath, ops=oracle_ops)
        sd = m.state_dict()
        for k, v in tgt.items():
            assert torch.equal(sd[k], v)
        for k, v in drf.items():
            assert torch.equal(sd["glide." + k], v)
    out = tmp_path / "resaved"
    checkpoint.save_draft_checkpoint(m.glide, str(out), cfg2)
    again = checkpoint.read_state_dict(str(out))
    assert sorted(again) == sorted(checkpoint.DRAFT_TENSORS)


def test_partial_hub_snapshot_falls_through_to_the_download(tmp_path, monkeypatch):
    """`AutoConfig.from_pretrained(hub_id)` leaves a config-only snapshot in the HF cache; the local-only pass
    finds it and must NOT be taken for the checkpoint -- the networked pass has to run (and fetch the tensors)."""
    import json
    import huggingface_hub
    from safetensors.torch import save_file
    from longspec_amd import checkpoint
    partial, full = tmp_path / "partial", tmp_path / "full"
    os.makedirs(partial), os.makedirs(full)
    for d in (partial, full):
        json.dump({"hidden_size": 256, "num_attention_heads": 2}, open(d / "config.json", "w"))
    save_file({"w": torch.ones(2)}, str(full / "model.safetensors"))
    calls = []

    def fake(repo, local_files_only=False, allow_patterns=None):
        calls.append(local_files_only)
        return str(partial if local_files_only else full)

    monkeypatch.setattr(huggingface_hub, "snapshot_download", fake)
    assert checkpoint.resolve_path("org/some-model") == str(full) and calls == [True, False]
    assert list(checkpoint.read_state_dict("org/some-model")) == ["w"]

    def offline(repo, local_files_only=False, allow_patterns=None):
        if not local_files_only:
            raise OSError("no network")
        return str(partial)

    monkeypatch.setattr(huggingface_hub, "snapshot_download", offline)
    assert checkpoint.load_config("org/some-model").hidden_size == 256        # the config alone is enough for load_config
    try:
        checkpoint.read_state_dict("org/some-model")
        raise AssertionError("a config-only snapshot must not pass for a checkpoint")
    except FileNotFoundError as e:
        assert "partial snapshot" in str(e)
    # an index naming an absent shard is just as incomplete
    json.dump({"weight_map": {"w": "model-00001-of-00002.safetensors"}}, open(partial / "model.safetensors.index.json", "w"))
    monkeypatch.setattr(huggingface_hub, "snapshot_download", fake)
    assert checkpoint.resolve_path("org/some-model") == str(full)
    # ... and so is a directory of numbered shards without an index that lacks some of them (an interrupted download)
    os.remove(partial / "model.safetensors.index.json")
    save_file({"w": torch.ones(2)}, str(partial / "model-00001-of-00003.safetensors"))
    calls.clear()
    assert checkpoint.resolve_path("org/some-model") == str(full) and calls == [True, False]
    # load_config on a config-only snapshot reads the cached config.json WITHOUT the networked pass
    calls.clear()
    assert checkpoint.load_config("org/some-model").hidden_size == 256 and calls == [True]
