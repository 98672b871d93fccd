"""ls_linear_fwd (weight-streaming skinny GEMM, packed weights) against the oracle's exact linear.

Tolerance: the kernel accumulates in fp32 (MFMA) in a fixed order and rounds once, the oracle rounds the
exact fp64 dot product once.  The fp32 accumulation error is bounded by a few 2^-24 of sum_k |x_k w_k|
(it matters only where the dot product cancels to ~0); on top of it the two roundings may land on
neighbouring values where the sum is a near-tie.  So: |got - want| <= 1 ulp(want) + 4 * 2^-24 * |x| . |w|^T,
and equality on >= 99 % of the elements."""
import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu


def _ulp(t):
    """One unit in the last place of each element of a half/bfloat16 tensor (as fp64)."""
    a = t.double().abs().clamp_min(6.1e-5 if t.dtype == torch.float16 else 1e-30)
    mant = 10 if t.dtype == torch.float16 else 7
    return torch.exp2(torch.floor(torch.log2(a)) - mant)


def _acc_tol(x, w):
    return 4.0 * 2.0 ** -24 * (x.double().abs().reshape(-1, x.shape[-1]) @ w.double().abs().t())


def _check(got, want, acc_tol=0.0):
    diff = (got.double() - want.double()).abs().reshape(-1, want.shape[-1])
    bound = _ulp(want).reshape(-1, want.shape[-1]) * 1.001 + acc_tol
    assert bool((diff <= bound).all()), f"more than 1 ulp: worst excess {(diff - bound).max().item():.3e}"
    assert (got == want).double().mean().item() >= 0.99


def _mk(shape, seed, scale=1.0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


SHAPES = [  # (N, K): Llama-3-8B projections (q/o, k/v, down), toy sizes, ragged N
    (4096, 4096), (1024, 4096), (4096, 14336), (512, 256), (512, 896), (100, 192), (264, 640),
    # cfg4 LongChat-13B (hidden 5120, intermediate 13824): q/k/v/o, gate/up rows, down_proj
    (5120, 5120), (13824, 5120), (5120, 13824),
    # cfg5 QwQ-32B (hidden 5120, kv 1024, intermediate 27648): k/v, gate/up rows, down_proj
    (1024, 5120), (27648, 5120), (5120, 27648),
]


@pytest.mark.parametrize("N,K", SHAPES)
@pytest.mark.parametrize("M", [1, 5, 16, 17, 32, 33, 74, 80])
def test_linear_matches_oracle(N, K, M):
    from longspec_amd import ops
    w, x = _mk((N, K), N + K, 0.03), _mk((M, K), M + K)
    b = _mk((N,), 7, 0.1)
    pw = ops.pack_weight(w.cuda())
    _check(ops.linear(x.cuda(), pw).cpu(), ref_ops.linear(x, w), _acc_tol(x, w))
    _check(ops.linear(x.cuda(), pw, b.cuda()).cpu(), ref_ops.linear(x, w, b), _acc_tol(x, w))


@pytest.mark.parametrize("S", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("M", [1, 74])
def test_split_k_is_exact_and_deterministic(S, M):
    from longspec_amd import ops
    N, K = 1024, 4096
    w, x = _mk((N, K), 11, 0.03), _mk((M, K), 12)
    pw = ops.pack_weight(w.cuda())
    y0 = ops.linear(x.cuda(), pw, n_splits=S)
    _check(y0.cpu(), ref_ops.linear(x, w), _acc_tol(x, w))
    for _ in range(5):                                   # arrival order of the workgroups must not matter
        assert torch.equal(ops.linear(x.cuda(), pw, n_splits=S), y0)


@pytest.mark.parametrize("N,K", [(152064, 5120), (32000, 5120)])
@pytest.mark.parametrize("M", [1, 74])
def test_lm_head_shapes_cfg4_cfg5(N, K, M):
    """lm_head of QwQ-32B (cfg5: 152064 x 5120) and LongChat-13B (cfg4: 32000 x 5120) against the oracle's exact linear,
    bf16 for the QwQ shape (how inference_qwq.py runs it).  The oracle evaluates 2000 sampled output columns (the full
    152064-column product in fp64 on the host would take minutes)."""
    from longspec_amd import ops
    dtype = torch.bfloat16 if N == 152064 else torch.float16
    w, x = _mk((N, K), N + K, 0.03, dtype), _mk((M, K), M + K, 1.0, dtype)
    y = ops.linear(x.cuda(), ops.pack_weight(w.cuda())).cpu()
    cols = torch.randperm(N, generator=torch.Generator().manual_seed(5))[:2000]
    _check(y[:, cols], ref_ops.linear(x, w[cols]), _acc_tol(x, w[cols]))


def test_split_k_under_concurrent_hbm_load():
    """The last-arriver split-K reduction must see every partial although the chip is busy: a side stream keeps
    HBM and the fabric loaded (large copies) while split-K GEMMs of every projection shape run back to back; each
    result must equal, bit for bit, the one computed on the idle chip.  (Round 1's workgroup-scope fence let the
    slab counter overtake partial stores still in flight.)"""
    from longspec_amd import ops
    side = torch.cuda.Stream()
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")        # 1 GiB
    big2 = torch.empty_like(big)
    cases = []
    for (N, K, M, S) in [(4096, 4096, 74, 0), (1024, 4096, 74, 8), (4096, 14336, 74, 0), (4096, 4096, 1, 0),
                         (5120, 5120, 74, 0), (1024, 4096, 16, 5), (5120, 13824, 74, 0)]:
        w, x = _mk((N, K), N + K + M, 0.03).cuda(), _mk((M, K), M + K + 1).cuda()
        pw = ops.pack_weight(w)
        kw = {"n_splits": S} if S else {}
        cases.append((x, pw, kw, ops.linear(x, pw, **kw).clone()))
    torch.cuda.synchronize()
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(6):
                big2.copy_(big, non_blocking=True)
        outs = []
        for _ in range(40):
            for (x, pw, kw, _want) in cases:
                outs.append(ops.linear(x, pw, **kw))
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, cases[i % len(cases)][3]), f"split-K result changed under load (rep {rep}, launch {i})"


def test_rows_do_not_depend_on_the_batch():
    """A token's projection is bit-identical whether it is computed alone (vanilla decode) or as one of 74
    verification rows: the k order of the accumulation depends on (N, K) only."""
    from longspec_amd import ops
    for (N, K) in [(4096, 4096), (14336, 4096), (4096, 14336), (512, 256)]:
        w, x = _mk((N, K), 21, 0.03), _mk((74, K), 22)
        pw = ops.pack_weight(w.cuda())
        full = ops.linear(x.cuda(), pw)
        for rows in (slice(0, 1), slice(5, 21), slice(40, 72)):
            assert torch.equal(ops.linear(x[rows].cuda(), pw), full[rows])


@pytest.mark.parametrize("M", [1, 16, 30, 74])
def test_qkv_in_one_launch(M):
    from longspec_amd import ops
    K = 4096
    ws = [_mk((n, K), 30 + i, 0.03) for i, n in enumerate((4096, 1024, 1024))]
    bs = [_mk((n,), 40 + i, 0.1) for i, n in enumerate((4096, 1024, 1024))]
    x = _mk((1, M, K), 50)
    outs = ops.linear_multi(x.cuda(), [ops.pack_weight(w.cuda()) for w in ws], [b.cuda() for b in bs])
    for o, w, b in zip(outs, ws, bs):
        assert o.shape == (1, M, w.shape[0])
        _check(o.cpu(), ref_ops.linear(x, w, b), _acc_tol(x, w))


@pytest.mark.parametrize("N,K", [(4096, 14336), (4096, 4096), (1000, 256), (128256, 512)])
@pytest.mark.parametrize("M", [1, 16, 33, 74])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_residual_epilogue(N, K, M, dtype):
    """`residual + linear(x)` in the epilogue == the projection followed by torch's add, bit for bit."""
    from longspec_amd import ops
    x = _mk((1, M, K), 3, dtype=dtype).cuda()
    w = ops.pack_weight(_mk((N, K), 4, K ** -0.5, dtype=dtype).cuda())
    b = _mk((N,), 5, 0.3, dtype=dtype).cuda()
    r = _mk((1, M, N), 6, dtype=dtype).cuda()
    for bias in (None, b):
        assert torch.equal(ops.linear(x, w, bias, residual=r), r + ops.linear(x, w, bias))
    rs = _mk((1, M, N + 8), 7, dtype=dtype).cuda()[..., :N]            # strided residual rows
    assert torch.equal(ops.linear(x, w, None, residual=rs), rs + ops.linear(x, w, None))
    with pytest.raises(ValueError):
        ops.linear(x, w, None, residual=r[..., :N - 4])


@pytest.mark.parametrize("dims", [(4096, 1024, 4096), (512, 128, 256), (5120, 5120, 5120), (1024, 256, 896)],
                         ids=lambda d: "x".join(map(str, d)))
@pytest.mark.parametrize("M", [1, 6, 16, 30, 74, 80])
@pytest.mark.parametrize("dtype,bias", [(torch.float16, False), (torch.float16, True), (torch.bfloat16, True)],
                         ids=["f16", "f16_bias", "bf16_bias"])
def test_qkv_rope_in_one_launch(dims, M, dtype, bias):
    """q|k|v projection with apply_rotary_pos_emb in the epilogue == the unfused launch followed by rope_apply_,
    bit for bit (same accumulation order, same roundings); the unfused pair is checked against the oracle above
    and in test_gpu_ops."""
    from longspec_amd import ops
    Nq, Nkv, K = dims
    x = _mk((1, M, K), 1, dtype=dtype).cuda()
    ws = [_mk((n, K), 2 + i, K ** -0.5, dtype=dtype).cuda() for i, n in enumerate((Nq, Nkv, Nkv))]
    bs = [_mk((n,), 7 + i, 0.5, dtype=dtype).cuda() if bias else None for i, n in enumerate((Nq, Nkv, Nkv))]
    pos = torch.arange(1000, 1000 + M)[None]
    inv_freq = 1.0 / (10000 ** (torch.arange(0, 128, 2).float() / 128))
    cos, sin = ops.rope_cos_sin(pos.cuda(), inv_freq.cuda(), 1.0, dtype)
    q0, k0, v0 = ops.linear_multi(x, [ops.pack_weight(w) for w in ws], bs)
    q0 = q0.reshape(1, M, Nq // 128, 128).clone()
    k0 = k0.reshape(1, M, Nkv // 128, 128).clone()
    ops.rope_apply_(q0, k0, cos, sin)
    q1, k1, v1 = ops.linear_qkv_rope(x, [ops.pack_weight(w, rope=i < 2) for i, w in enumerate(ws)], bs, cos, sin)
    assert torch.equal(q1.reshape(q0.shape), q0) and torch.equal(k1.reshape(k0.shape), k0) and torch.equal(v1, v0)
    # q alone (the draft's cross-attention); its split-K factor is that of the q weight alone
    q3 = ops.linear(x, ops.pack_weight(ws[0]), bs[0]).reshape(q0.shape).clone()
    ops.rope_apply_(q3, q3[:, :, :0], cos, sin)
    (q2,) = ops.linear_qkv_rope(x, [ops.pack_weight(ws[0], rope=True)], bs[:1], cos, sin)
    assert torch.equal(q2.reshape(q0.shape), q3)
    # the layouts are not interchangeable
    with pytest.raises(ValueError):
        ops.linear_qkv_rope(x, [ops.pack_weight(w) for w in ws], bs, cos, sin)
    with pytest.raises(ValueError):
        ops.linear_multi(x, [ops.pack_weight(w, rope=True) for w in ws], bs)


@pytest.mark.parametrize("N,K", [(14336, 4096), (512, 256), (1024, 896), (1536, 512), (13824, 5120), (27648, 5120)])
@pytest.mark.parametrize("M", [1, 16, 30, 74])
def test_mlp_gate_up_silu(N, K, M):
    """silu(gate) * up with the reference's rounding points.  Each inner GEMM may be 1 ulp off the exact rounding on a
    rare element.  First-order propagation of those two input perturbations through out = round(round(silu(g)) * u):
        |d out| <= |u| |silu'(g)| ulp(g) + |silu(g)| ulp(u)  +  the two output-side roundings (<= 2 ulp(out)),
    which is <= 4 ulp(out) for g > -2 but MORE for strongly negative g, where silu is ill-conditioned in relative terms
    (g = -4.35: |g silu'(g) / silu(g)| = 3.3, so one ulp of g moves the output by up to ~6 of its ulps -- met once in the
    2 M outputs of the QwQ-32B shape).  The bound below is that sum; >= 98 % of the elements must be equal."""
    from longspec_amd import ops
    wg, wu, x = _mk((N, K), 61, 0.03), _mk((N, K), 62, 0.03), _mk((M, K), 63)
    got = ops.mlp_gate_up(x.cuda(), ops.pack_gate_up(wg.cuda(), wu.cuda())).cpu()
    want = ref_ops.mlp_gate_up(x, wg, wu)
    gt, ut = ref_ops.linear(x, wg), ref_ops.linear(x, wu)                 # the exactly rounded projections
    g64, u64 = gt.double(), ut.double()
    sig = torch.sigmoid(g64)
    dsilu = sig * (1.0 + g64 * (1.0 - sig))
    bound = (u64.abs() * dsilu.abs() * _ulp(gt) + (g64 * sig).abs() * _ulp(ut)) * 1.001 + 2.001 * _ulp(want) + 1e-4
    diff = (got.double() - want.double()).abs()
    assert bool((diff <= bound).all()), f"max excess {(diff - bound).max().item():.3e}"
    assert bool((diff <= 8.001 * _ulp(want) + 1e-4).all()), f"max diff {diff.max().item():.3e}"
    assert (got == want).double().mean().item() >= 0.98


def test_bf16_and_strided_input():
    from longspec_amd import ops
    N, K, M = 1024, 512, 20
    w, xbig = _mk((N, K), 71, 0.03, torch.bfloat16), _mk((M, 2 * K), 72, 1.0, torch.bfloat16)
    x = xbig[:, K:]                                       # row stride 2K, 16-byte aligned start
    _check(ops.linear(x.cuda(), ops.pack_weight(w.cuda())).cpu(), ref_ops.linear(x, w), _acc_tol(x, w))


def test_unsupported_shapes_fail_loudly():
    from longspec_amd import ops
    from longspec_amd._C import LongSpecHipError
    w = _mk((256, 256), 81).cuda()
    pw = ops.pack_weight(w)
    with pytest.raises(LongSpecHipError, match="plain library GEMM"):
        ops.linear(torch.zeros(81, 256, dtype=torch.float16, device="cuda"), pw)
    assert not ops.linear_supported(torch.zeros(81, 256, dtype=torch.float16, device="cuda"), 256)
    assert not ops.linear_supported(torch.zeros(4, 160, dtype=torch.float16, device="cuda"), 160)
    with pytest.raises(TypeError, match="PackedWeight"):
        ops.linear(torch.zeros(4, 256, dtype=torch.float16, device="cuda"), w)


def test_decode_linear_module_repacks_after_weight_update():
    from longspec_amd import ops
    from longspec_amd.llama import DecodeLinear
    lin = DecodeLinear(256, 512, bias=True, ops=ops).half().cuda()
    x = _mk((3, 256), 91).cuda()
    y0 = lin(x)
    _check(y0.cpu(), ref_ops.linear(x.cpu(), lin.weight.detach().cpu(), lin.bias.detach().cpu()), 1e-5)
    with torch.no_grad():
        lin.weight.mul_(2.0)
    _check(lin(x).cpu(), ref_ops.linear(x.cpu(), lin.weight.detach().cpu(), lin.bias.detach().cpu()), 1e-5)
    big = _mk((200, 256), 92).cuda()                      # prefill-shaped: library GEMM
    assert lin(big).shape == (200, 512)


@pytest.mark.parametrize("hidden,inter", [(4096, 14336), (5120, 13824), (256, 512), (896, 1024)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("M", [1, 7, 16, 30, 74, 80])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_rmsnorm_folded_into_the_projections(hidden, inter, M, dtype):
    """The producer side (`ssq_out`: the partial sums of squares of what o_proj / down_proj store, residual included) and
    the consumer side (`norm=NormFold`: q|k|v with the rotary epilogue, gate|up + SiLU, a plain projection) of the folded
    LlamaRMSNorm against the chain it replaces -- `rmsnorm(residual + y)` then the plain launch --, bit for bit: the sum of
    squares is taken in one canonical order by both (csrc/ls_common.h)."""
    from longspec_amd import ops
    eps = 1e-5
    x = _mk((1, M, inter), 1, dtype=dtype).cuda()
    res = _mk((1, M, hidden), 2, 3.0, dtype=dtype).cuda()
    w_down = ops.pack_weight(_mk((hidden, inter), 3, inter ** -0.5, dtype=dtype).cuda())
    nw = (1.0 + 0.1 * _mk((hidden,), 4)).to(dtype).cuda()
    # producer: residual + down_proj(x) with the row statistics on the side
    h_ref = ops.linear(x, w_down, None, residual=res)
    h, ssq = ops.linear(x, w_down, None, residual=res, ssq_out=True)
    assert torch.equal(h, h_ref) and ssq.shape == (M, hidden // 64)
    want_ssq = h.float().reshape(M, hidden // 64, 64).pow(2).sum(-1)
    assert torch.allclose(ssq, want_ssq, rtol=1e-5, atol=0)
    hn = ops.rmsnorm(h, nw, eps)                                   # the stand-alone kernel, same canonical order
    fold = ops.NormFold(nw, eps, ssq)
    # consumer 1: gate|up + SiLU
    gu = ops.pack_gate_up(_mk((inter, hidden), 5, hidden ** -0.5, dtype=dtype).cuda(),
                          _mk((inter, hidden), 6, hidden ** -0.5, dtype=dtype).cuda())
    assert torch.equal(ops.mlp_gate_up(h, gu, norm=fold), ops.mlp_gate_up(hn, gu))
    # consumer 2: a plain projection with bias (lm_head-like, ragged N), and again as a producer
    N2 = 1000 if hidden < 1024 else 3000
    w2 = ops.pack_weight(_mk((N2, hidden), 7, hidden ** -0.5, dtype=dtype).cuda())
    b2 = _mk((N2,), 8, 0.3, dtype=dtype).cuda()
    assert torch.equal(ops.linear(h, w2, b2, norm=fold), ops.linear(hn, w2, b2))
    # consumer 3: q|k|v with the rotary epilogue (heads x 128 columns)
    if hidden % 128 == 0:
        Nq, Nkv = hidden, max(128, hidden // 4 // 128 * 128)
        ws = [_mk((n, hidden), 9 + i, hidden ** -0.5, dtype=dtype).cuda() for i, n in enumerate((Nq, Nkv, Nkv))]
        bs = [_mk((n,), 12 + i, 0.5, dtype=dtype).cuda() for i, n in enumerate((Nq, Nkv, Nkv))]
        pos = torch.arange(500, 500 + M)[None]
        inv_freq = 1.0 / (10000 ** (torch.arange(0, 128, 2).float() / 128))
        cos, sin = ops.rope_cos_sin(pos.cuda(), inv_freq.cuda(), 1.0, dtype)
        packed = [ops.pack_weight(w, rope=i < 2) for i, w in enumerate(ws)]
        for a, b in zip(ops.linear_qkv_rope(h, packed, bs, cos, sin, norm=fold), ops.linear_qkv_rope(hn, packed, bs, cos, sin)):
            assert torch.equal(a, b)
        plain = [ops.pack_weight(w) for w in ws]
        for a, b in zip(ops.linear_multi(h, plain, bs, norm=fold), ops.linear_multi(hn, plain, bs)):
            assert torch.equal(a, b)
    # a second norm of the same stream through the residual path of the stand-alone kernel
    y2 = ops.linear(x, w_down, None)
    hn2, h2 = ops.rmsnorm(y2, nw, eps, residual=res)
    assert torch.equal(h2, h) and torch.equal(hn2, hn)
    with pytest.raises(ValueError):
        ops.mlp_gate_up(h, gu, norm=ops.NormFold(nw[:-8].contiguous(), eps, ssq))
