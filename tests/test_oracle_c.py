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
