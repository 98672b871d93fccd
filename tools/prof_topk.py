import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from longspec_amd import ops
g = torch.Generator().manual_seed(0)
for R, k in [(16, 16), (16, 1), (1, 4), (4, 16)]:
    x = (torch.randn(1, R, 128256, generator=g) * 2).half().cuda()
    h = torch.zeros(1, R).cuda()
    for _ in range(5):
        ops.logprob_topk(x, h, k)
x = (torch.randn(1, 69, 128256, generator=g) * 2).half().cuda()
for _ in range(5):
    ops.argmax_rows(x)
torch.cuda.synchronize()
