"""CPU-side checks of round-4 host logic that needs no GPU: the layer-tail planner (asked through the C ABI, which only looks
at shapes), bench.py's padded-row count and per-model defaults."""
import torch


def test_layer_tail_planner_takes_the_baseline_models_and_only_verification_sized_passes():
    from longspec_amd import ops
    old = ops.LAYER_TAIL
    ops.LAYER_TAIL = True
    try:
        dims = {"llama3-8b": (4096, 14336, 32, 8), "vicuna-7b": (4096, 11008, 32, 32), "longchat-13b": (5120, 13824, 40, 40),
                "qwq-32b": (5120, 27648, 40, 8), "toy": (256, 512, 2, 2)}
        for name, (hidden, inter, H, Hkv) in dims.items():
            nq = (H * 128, Hkv * 128, Hkv * 128)
            assert ops.layer_tail_supported(74, hidden, inter, torch.float16, Ko=H * 128, n_qkv=nq), name
            assert ops.layer_tail_supported(74, hidden, inter, torch.bfloat16, Ko=H * 128), name      # last layer: no q|k|v behind it
            for rows in (1, 4, 16, 32, 81):          # draft / vanilla passes and prefill keep the launch chain
                assert not ops.layer_tail_supported(rows, hidden, inter, torch.float16, Ko=H * 128, n_qkv=nq), (name, rows)
        assert not ops.layer_tail_supported(74, 4096, 14336, torch.float32, Ko=4096)
        ops.LAYER_TAIL = False                       # the default: nothing takes the fused path
        assert not ops.layer_tail_supported(74, 4096, 14336, torch.float16, Ko=4096)
    finally:
        ops.LAYER_TAIL = old


def test_bench_issued_rows_and_defaults():
    import bench
    assert bench.issued_rows(296) == 320           # Llama-3 GQA-4: 19 tiles -> 4 pairs x 5
    assert bench.issued_rows(370) == 384           # QwQ GQA-5: 24 tiles as two chunks of 12
    assert bench.issued_rows(74) == 80             # MHA: the general kernel, 16-row tiles
    assert [c["model"] for c in bench.BASELINE_CONFIGS] == ["vicuna-7b-16k", "llama3-8b-262k", "llama3-8b-262k", "longchat-13b-16k", "qwq-32b"]
