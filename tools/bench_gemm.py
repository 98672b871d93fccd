"""Microbenchmark + check of the skinny linear kernel (ls_linear_fwd) against torch (hipBLASLt).

    python tools/bench_gemm.py [--rows 74] [--splits 0]
"""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from longspec_amd import ops


def timeit(fn, n=40):
    """Mean GPU time per call: one event pair around EACH call (so CPU launch overhead, which exceeds
    the duration of the small kernels, is not counted), median of n."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="*", default=[74, 16, 1])
    ap.add_argument("--splits", type=int, nargs="*", default=[0])
    args = ap.parse_args()
    g = torch.Generator(device="cpu").manual_seed(0)
    shapes = [("q_proj", 4096, 4096), ("kv_proj", 1024, 4096), ("gate/up", 14336, 4096), ("down", 4096, 14336),
              ("lm_head", 128256, 4096)]
    nrot = 6          # rotate over copies of W so that every call streams from HBM, not from the 256 MB MALL
    for M in args.rows:
        for name, N, K in shapes:
            Ws = [(torch.randn(N, K, generator=g) * 0.02).half().cuda() for _ in range(nrot if N * K * 2 < 300e6 else 2)]
            PW = [ops.pack_weight(w) for w in Ws]
            x = (torch.randn(M, K, generator=g)).half().cuda()
            ref = (x.double() @ Ws[0].double().t())
            for S in args.splits:
                y = ops.linear(x, PW[0], n_splits=S)
                err = (y.double() - ref).abs().max().item()
                refh = torch.nn.functional.linear(x, Ws[0])
                err_t = (refh.double() - ref).abs().max().item()
                i = [0]

                def f_ours():
                    i[0] = (i[0] + 1) % len(Ws)
                    return ops.linear(x, PW[i[0]], n_splits=S)

                def f_torch():
                    i[0] = (i[0] + 1) % len(Ws)
                    return torch.nn.functional.linear(x, Ws[i[0]])
                t_o, t_t = timeit(f_ours), timeit(f_torch)
                gb = N * K * 2 / 1e9
                print(f"M={M:3d} {name:8s} N={N:6d} K={K:5d} S={S}: ours {t_o:7.1f} us ({gb / t_o * 1e6 / 1e3:5.2f} TB/s) "
                      f"torch {t_t:7.1f} us ({gb / t_t * 1e6 / 1e3:5.2f} TB/s)  maxerr ours {err:.2e} torch {err_t:.2e}", flush=True)
            del Ws, PW
        if M <= 96:
            Wg = [(torch.randn(14336, 4096, generator=g) * 0.02).half().cuda() for _ in range(3)]
            Wu = [(torch.randn(14336, 4096, generator=g) * 0.02).half().cuda() for _ in range(3)]
            PWgu = [ops.pack_gate_up(g_, u_) for g_, u_ in zip(Wg, Wu)]
            x = (torch.randn(M, 4096, generator=g)).half().cuda()
            want = torch.nn.functional.silu(torch.nn.functional.linear(x, Wg[0])) * torch.nn.functional.linear(x, Wu[0])
            got = ops.mlp_gate_up(x, PWgu[0])
            i = [0]

            def f1():
                i[0] = (i[0] + 1) % 3
                return ops.mlp_gate_up(x, PWgu[i[0]])

            def f2():
                i[0] = (i[0] + 1) % 3
                return torch.nn.functional.silu(torch.nn.functional.linear(x, Wg[i[0]])) * torch.nn.functional.linear(x, Wu[i[0]])
            t1, t2 = timeit(f1), timeit(f2)
            gb = 2 * 14336 * 4096 * 2 / 1e9
            print(f"M={M:3d} gate_up_silu: ours {t1:7.1f} us ({gb / t1 * 1e3:5.2f} TB/s) torch {t2:7.1f} us ({gb / t2 * 1e3:5.2f} TB/s) "
                  f"maxdiff {(got.float() - want.float()).abs().max().item():.2e} nonequal {(got != want).float().mean().item():.4f}", flush=True)
            Wq = [(torch.randn(n, 4096, generator=g) * 0.02).half().cuda() for n in (4096, 1024, 1024)]
            bq = [(torch.randn(n, generator=g) * 0.02).half().cuda() for n in (4096, 1024, 1024)]
            PWq = [ops.pack_weight(w) for w in Wq]
            outs = ops.linear_multi(x, PWq, bq)
            for o, w, b in zip(outs, PWq, bq):
                assert (o.float() - ops.linear(x, w, b).float()).abs().max().item() < 4e-3, "multi != single"
            t3 = timeit(lambda: ops.linear_multi(x, PWq, bq))
            t4 = timeit(lambda: [torch.nn.functional.linear(x, w, b) for w, b in zip(Wq, bq)])
            print(f"M={M:3d} qkv fused (MALL-resident): ours {t3:7.1f} us torch 3 calls {t4:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
