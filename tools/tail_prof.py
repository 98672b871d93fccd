"""Wall-clock timeline of one layer-tail launch (round 4): every workgroup stamps s_memrealtime (100 MHz) at its phase
boundaries; this prints, per stamp, when the first / median / last workgroup got there (us from the first workgroup's entry).

    python tools/build_variant.py tailprof --src gemm.hip -DLS_TAIL_PROF
    LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_tailprof.so python tools/tail_prof.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from longspec_amd import ops
import test_gpu_tail as T

NAMES = ["entry", "P1 o_proj done", "arrived C1", "[norm] C1 seen", "[norm] N1 done + arrived C2", "C2 seen", "P2 gate|up done",
         "arrived C3", "C3 seen", "P3 down done", "arrived C4", "[norm] C4 seen", "[norm] N2 done + arrived C5", "C5 seen", "P4 q|k|v done"]


def main():
    hidden, inter, H, Hkv, bias = T.DIMS["llama3-8b"]
    M, dtype, eps, dev = 74, torch.float16, 1e-5, "cuda"
    g = torch.Generator().manual_seed(5)
    Ws = [T._weights(g, hidden, inter, H, Hkv, bias, dtype) for _ in range(3)]      # rotate: nothing served from the MALL
    pos = torch.arange(5000, 5000 + M, dtype=torch.int64, device=dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128)).to(dev)
    cos, sin = ops.rope_cos_sin(pos[None], inv, 1.0, dtype)
    cos, sin = cos.reshape(M, 128).contiguous(), sin.reshape(M, 128).contiguous()
    attn = torch.randn(M, H * 128, generator=g).to(dtype).to(dev)
    resid = torch.randn(M, hidden, generator=g).to(dtype).to(dev)
    rows = []
    for it in range(9):
        W = Ws[it % 3]
        ops.layer_tail(attn, resid.clone(), W["o"], W["n1"], W["gu"], W["d"], W["n2"], eps, qkv_weights=W["qkv"], qkv_biases=W["bq"],
                       cos=cos, sin=sin)
        torch.cuda.synchronize()
        ws = ops._tail_ws[torch.cuda.current_device()]
        need = list(ops._tail_need.values())[0]
        G = torch.cuda.get_device_properties(0).multi_processor_count
        st = ws[need - G * 128:need].view(torch.int64).view(G, 16).cpu().double() / 100.0       # us
        if it >= 3:
            rows.append(st)
    st = torch.stack(rows).median(dim=0).values
    t0 = st[:, 0].min()
    out = {}
    for i, nm in enumerate(NAMES):
        col = st[:, i]
        # (the stamps of a norm phase exist only in the workgroups that ran it: the others hold zeros)
        sel = (col > 0) if nm.startswith("[norm]") else torch.ones(G, dtype=torch.bool)
        if i in (5, 6, 7):
            sel = sel & (torch.arange(G) < 224)
        if i in (13, 14):
            sel = sel & (torch.arange(G) < 192)
        c = col[sel] - t0
        out[nm] = {"first": round(c.min().item(), 2), "median": round(c.median().item(), 2), "last": round(c.max().item(), 2)}
        print(f"{nm:32s} first {c.min().item():7.2f}  median {c.median().item():7.2f}  last {c.max().item():7.2f} us")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
