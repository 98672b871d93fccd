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
