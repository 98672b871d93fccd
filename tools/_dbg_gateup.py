import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from longspec_amd import ops
from oracle import ref_ops
def _mk(shape, seed, scale=1.0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)
N, K, M = 27648, 5120, 74
wg, wu, x = _mk((N, K), 61, 0.03), _mk((N, K), 62, 0.03), _mk((M, K), 63)
got = ops.mlp_gate_up(x.cuda(), ops.pack_gate_up(wg.cuda(), wu.cuda())).cpu()
want = ref_ops.mlp_gate_up(x, wg, wu)
a = want.double().abs().clamp_min(6.1e-5)
ulp = torch.exp2(torch.floor(torch.log2(a)) - 10)
d = (got.double() - want.double()).abs()
r = d / ulp
idx = torch.nonzero(d > 4.001 * ulp + 1e-4)
print("violations", idx.shape[0], "of", d.numel())
g32 = (x.float() @ wg.float().t()); u32 = (x.float() @ wu.float().t())
for i, j in idx[:12].tolist():
    print(i, j, "got", got[i, j].item(), "want", want[i, j].item(), "ulps", r[i, j].item(), "g", g32[i, j].item(), "u", u32[i, j].item())
# also the plain linears for the same weights
yg = ops.linear(x.cuda(), ops.pack_weight(wg.cuda())).cpu()
print("gate linear max ulp diff vs fp32 ref rounded:", ((yg.double() - g32.half().double()).abs() / torch.exp2(torch.floor(torch.log2(g32.double().abs().clamp_min(6.1e-5))) - 10)).max().item())
