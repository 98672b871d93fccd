"""The C restatement (cpu_baseline port) against the Python oracle and the reference's goldens."""
import pytest
import torch

import cases
import toy
from oracle import c_port, ref_ops


@pytest.mark.parametrize("c", list(cases.verify_cases()), ids=lambda c: c["name"])
@pytest.mark.parametrize("last_layer", [False, True])
def test_c_port_matches_reference_golden(c, last_layer):
    kc, vc = c["kc"].clone(), c["vc"].clone()
    out = c_port.verify_attention(c["q"], c["k"], c["v"], kc, vc, c["L"], c["mask"], last_layer)
    d = (out.float() - c["hybrid"][last_layer].float()).abs()
    assert d.max().item() <= 1.1e-3 and (d > 0).float().mean().item() < 0.02
    L = c["L"]
    assert torch.equal(kc[:, L:L + 74], c["k"]) and torch.equal(vc[:, L:L + 74], c["v"])


def test_c_port_matches_python_oracle_gqa():
    q, k, v, kc, vc, tm = toy.verify_inputs(8, 2, 777, 99, a=5)
    cl = torch.tensor([777], dtype=torch.int32)
    ref = ref_ops.target_verify_attention(q, k, v, kc.clone(), vc.clone(), cl, tm, False)
    out = c_port.verify_attention(q, k, v, kc, vc, 777, tm, False)
    d = (out.float() - ref.float()).abs()
    assert d.max().item() <= 1.1e-3 and (d > 0).float().mean().item() < 0.02


@pytest.mark.parametrize("last_layer", [False, True])
def test_c_port_bf16_matches_python_oracle(last_layer):
    """The bfloat16 form of the C restatement (cfg5: QwQ-32B runs in bf16; tests/test_gpu_ops.py uses it at 32k) against the
    Python oracle evaluated in bf16: the same rounding points, one bf16 ulp on a sliver of elements (fp32 summation order)."""
    q, k, v, kc, vc, tm = toy.verify_inputs(10, 2, 333, 5, a=3)
    q, k, v, kc, vc = (t.to(torch.bfloat16) for t in (q, k, v, kc, vc))
    cl = torch.tensor([333], dtype=torch.int32)
    ref = ref_ops.target_verify_attention(q, k, v, kc.clone(), vc.clone(), cl, tm, last_layer)
    out = c_port.verify_attention(q, k, v, kc, vc, 333, tm, last_layer)
    assert out.dtype == torch.bfloat16
    d = (out.float() - ref.float()).abs()
    assert d.max().item() <= 2.0 ** -8 * ref.float().abs().max().item() and (d > 0).float().mean().item() < 0.02
    assert torch.equal(kc[:, 333:333 + 74], k) and torch.equal(vc[:, 333:333 + 74], v)
