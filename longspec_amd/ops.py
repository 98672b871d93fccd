"""Operator layer: torch tensors in, HIP kernels (through the C ABI) out.

One function per reference seam of the draft-then-verify round:

  kvcache_attention      flash_attn_with_kvcache   longspec/test/llama.py:324,385, llama_glide.py:261,265,297,300
  verify_attention       LlamaAttention.tree_decoding + tree_part_fwd + merge   llama.py:385-387,394-421
  draft_tree_attention   GlideAttention.tree_decoding (self-attn) + triton_tree_part_fwd  llama_glide.py:300-329
  tree_attention         triton_tree_attn.attention   triton_tree_attn.py:19-77
  rmsnorm / rope_*       LlamaRMSNorm / LlamaRotaryEmbedding / apply_rotary_pos_emb (transformers)
  tree_collapse          LlamaGlide.tree_verification  llama_glide.py:1128-1175
  lse_merge              N-way form of llama.py:385-387,420 for sequence-sharded prefix KV
  logprob_topk / argmax_rows   beam growth and greedy verification on the lm_head logits  llama_glide.py:1019-1064,1091
  linear / linear_multi / mlp_gate_up   the projections of a decode pass (M <= 80 token rows):
                         q/k/v/o_proj llama.py:361-363,390, LlamaMLP (qwen2.py:218-230), lm_head llama_glide.py:1091

Everything runs on the CURRENT torch stream, without host synchronisation.  There is
no fallback: a missing extension or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Optional, Tuple

import torch

from . import _C
from ._C import LS_F16, LS_NEW_DRAFT, LS_NEW_FLASH, LS_NEW_NONE, LS_NEW_TARGET, AttnDesc

_DT = {torch.float16: _C.LS_F16, torch.bfloat16: _C.LS_BF16}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """Raw hipStream_t of torch's current stream (the C accessors when this torch has them: the public
    torch.cuda.current_stream() costs ~8 us of Python per call, twice per operator)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _dtype(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype} (fp16/bf16 only)") from None


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("longspec_amd ops run on the GPU only (no CPU path): got a CPU tensor")


class _Workspace:
    """Per-device scratch for split-KV partials, grown on demand.  Calls on one stream are
    ordered, so one buffer per (device, stream) is reused by every call."""

    def __init__(self):
        self._buf = {}

    def get(self, device, nbytes: int) -> torch.Tensor:
        key = (device.index, _stream())
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf


_ws = _Workspace()
_causal_bits = {}


class _ZeroedWorkspace:
    """Scratch of the skinny-GEMM kernel (slab counters + split-K partials).  Zero-filled once when
    (re)allocated; every launch leaves its counters zero again (include/longspec_hip.h)."""

    def __init__(self):
        self._buf = {}

    def get(self, device, nbytes: int, stream: Optional[int] = None) -> torch.Tensor:
        key = (device.index, _stream() if stream is None else stream)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.zeros(max(nbytes, 32 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf


_gemm_ws = _ZeroedWorkspace()


def workspace_tensors():
    """The scratch buffers currently handed out (a captured HIP graph keeps them alive: it holds their addresses)."""
    return list(_ws._buf.values()) + list(_gemm_ws._buf.values())


_linear_need = {}            # (shape key) -> workspace bytes of the launch plan
LINEAR_MAX_ROWS = 80


def linear_supported(x: torch.Tensor, in_features: int) -> bool:
    """True when the weight-streaming kernel takes this shape; otherwise the caller has a plain library
    GEMM on its hands (prefill: M in the thousands), which is hipBLASLt's job, not this kernel's."""
    rows = x.numel() // x.shape[-1]
    return x.is_cuda and x.dtype in _DT and 1 <= rows <= LINEAR_MAX_ROWS and in_features % 64 == 0 and in_features >= 128


class PackedWeight:
    """An ``nn.Linear.weight`` [N, K] re-laid-out once by ``ls_linear_pack_weight`` into the MFMA A-operand
    order the skinny kernel streams (include/longspec_hip.h).  ``data`` is a flat tensor of the weight's
    dtype with ceil(N/16)*16*K elements."""
    __slots__ = ("data", "n", "k", "dtype", "rope")

    def __init__(self, data, n, k):
        self.data, self.n, self.k, self.dtype, self.rope = data, n, k, data.dtype, False


def pack_weight(weight: torch.Tensor, rope: bool = False) -> PackedWeight:
    """``rope=True``: a q_proj / k_proj weight [heads*128, K] for ``linear_qkv_rope`` (rotary pairs in neighbouring tiles)."""
    _dev(weight)
    if weight.dim() != 2 or weight.shape[1] % 32 != 0:
        raise ValueError("pack_weight: [N, K] weight with K a multiple of 32 expected")
    w = weight.detach().contiguous()
    N, K = w.shape
    lib = _C.load()
    nbytes = lib.ls_linear_packed_bytes(N, K)
    out = torch.empty(nbytes // 2, dtype=w.dtype, device=w.device)
    if rope:
        _C.check(lib.ls_linear_pack_rope(w.data_ptr(), out.data_ptr(), N, K, _dtype(w), _stream()), "ls_linear_pack_rope")
    else:
        _C.check(lib.ls_linear_pack_weight(w.data_ptr(), out.data_ptr(), N, K, _dtype(w), _stream()), "ls_linear_pack_weight")
    pw = PackedWeight(out, N, K)
    pw.rope = rope
    return pw


def pack_gate_up(gate_weight: torch.Tensor, up_weight: torch.Tensor) -> PackedWeight:
    """gate_proj / up_proj of one MLP as one packed matrix with alternating 16-row tiles (operand of mlp_gate_up)."""
    _dev(gate_weight, up_weight)
    if gate_weight.shape != up_weight.shape or gate_weight.dim() != 2 or gate_weight.shape[0] % 16 or gate_weight.shape[1] % 32:
        raise ValueError("pack_gate_up: two [N, K] weights with N % 16 == 0 and K % 32 == 0 expected")
    g, u = gate_weight.detach().contiguous(), up_weight.detach().contiguous()
    N, K = g.shape
    lib = _C.load()
    out = torch.empty(lib.ls_linear_packed_bytes(2 * N, K) // 2, dtype=g.dtype, device=g.device)
    _C.check(lib.ls_linear_pack_gate_up(g.data_ptr(), u.data_ptr(), out.data_ptr(), N, K, _dtype(g), _stream()),
             "ls_linear_pack_gate_up")
    return PackedWeight(out, N, K)


def logprob_topk(logits: torch.Tensor, history: Optional[torch.Tensor], k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(logits.float().log_softmax(-1) + history[..., None]).view(bsz, -1).topk(k)`` (llama_glide.py:1019-1020,
    1046-1064) straight from the fp16 lm_head output: logits [bsz, rows, V], history [bsz, rows] fp32 or None ->
    (values [bsz, k] fp32 descending, flat indices [bsz, k] int64 = row * V + column)."""
    _dev(logits, history)
    if logits.dim() == 2:
        logits = logits.unsqueeze(1)
    if logits.stride(-1) != 1 or logits.stride(1) % 8 != 0:
        logits = logits.contiguous()
    b, R, V = logits.shape
    if k > 64 or R > 128 or R * ((V + 8191) // 8192) * k > 5120 or R * ((V + 8191) // 8192) > 512 or V % 8:
        # beyond the fused kernel's candidate budget (trees wider than BASELINE's 16 per level): library kernels
        lp = logits.float().log_softmax(dim=-1)
        if history is not None:
            lp = lp + history[:, :, None].float()
        return lp.view(b, -1).topk(dim=-1, k=k, largest=True, sorted=True)
    vals = torch.empty((b, k), dtype=torch.float32, device=logits.device)
    idx = torch.empty((b, k), dtype=torch.int64, device=logits.device)
    if history is not None:
        history = history.to(torch.float32).contiguous()
    lib = _C.load()
    need = lib.ls_topk_workspace_bytes(R, V, k)
    ws = _ws.get(logits.device, max(need, 1))
    for i in range(b):
        _C.check(lib.ls_logprob_topk(logits[i].data_ptr(), R, V, logits.stride(1), _dtype(logits),
                                     history[i].data_ptr() if history is not None else None, k, vals[i].data_ptr(),
                                     idx[i].data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "ls_logprob_topk")
    return vals, idx


def argmax_rows(logits: torch.Tensor) -> torch.Tensor:
    """``logits.argmax(-1)`` for lm_head outputs [..., V] (first maximum), int64 [...]."""
    _dev(logits)
    shape = logits.shape[:-1]
    V = logits.shape[-1]
    x = logits.reshape(-1, V)
    if x.stride(-1) != 1 or x.stride(0) % 8 != 0:
        x = x.contiguous()
    out = torch.empty((x.shape[0],), dtype=torch.int64, device=logits.device)
    lib = _C.load()
    need = lib.ls_topk_workspace_bytes(x.shape[0], V, 1)
    ws = _ws.get(logits.device, max(need, 1))
    _C.check(lib.ls_argmax_rows(x.data_ptr(), x.shape[0], V, x.stride(0), _dtype(x), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                _stream()), "ls_argmax_rows")
    return out.view(shape)


TOPK_CHUNK = 8192          # logits per stage-1 record (ls_topk_chunk(); checked at first use)


def topk_stage1(logits_local: Optional[torch.Tensor], rows: int, k: int, col_base: int, nslots: int, out: torch.Tensor, dtype=None):
    """Stage 1 of the vocabulary-parallel ``logprob_topk`` / ``argmax_rows``: this rank's slice of the lm_head output
    ``logits_local`` [rows, V_local] (global columns ``col_base`` ...; None when the rank owns no column) -> ``nslots`` chunk
    records [nslots, rows, 2 + 2k] fp32 written into ``out`` (flat, e.g. the exchange's send buffer).  Returns the record view."""
    lib = _C.load()
    assert lib.ls_topk_chunk() == TOPK_CHUNK
    rec = 2 + 2 * k
    n = nslots * rows * rec
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= n
    if logits_local is None:
        v_local, ld, ptr = 0, 8, out.data_ptr()            # (never read: every slot is an empty record)
        dt = _C.LS_F16 if dtype in (None, torch.float16) else _C.LS_BF16
    else:
        _dev(logits_local)
        assert logits_local.dim() == 2 and logits_local.shape[0] == rows and logits_local.stride(1) == 1
        v_local, ld, ptr, dt = logits_local.shape[1], logits_local.stride(0), logits_local.data_ptr(), _dtype(logits_local)
    _C.check(lib.ls_topk_stage1(ptr, rows, v_local, ld, dt, k, col_base, nslots, out.data_ptr(), _stream()), "ls_topk_stage1")
    return out[:n].view(nslots, rows, rec)


def topk_stage2(records: torch.Tensor, rows: int, vocab: int, k: int, history: Optional[torch.Tensor], argmax: bool):
    """Stage 2: all ranks' records [slots, rows, 2 + 2k] in rank order (= global chunk order) -> what ``logprob_topk``
    (values [1, k], flat indices [1, k]) or ``argmax_rows`` (indices [rows]) return on the gathered logits, bit for bit."""
    _dev(records, history)
    assert records.dtype == torch.float32 and records.is_contiguous() and records.shape[1] == rows and records.shape[2] == 2 + 2 * k
    lib = _C.load()
    dev = records.device
    if argmax:
        idx = torch.empty((rows,), dtype=torch.int64, device=dev)
        _C.check(lib.ls_topk_stage2(records.data_ptr(), rows, vocab, 1, records.shape[0], None, 1, None, idx.data_ptr(), _stream()),
                 "ls_topk_stage2")
        return idx
    vals = torch.empty((1, k), dtype=torch.float32, device=dev)
    idx = torch.empty((1, k), dtype=torch.int64, device=dev)
    if history is not None:
        history = history.to(torch.float32).contiguous()
    _C.check(lib.ls_topk_stage2(records.data_ptr(), rows, vocab, k, records.shape[0], history.data_ptr() if history is not None else None,
                                0, vals.data_ptr(), idx.data_ptr(), _stream()), "ls_topk_stage2")
    return vals, idx


_linear_timing = None


def set_linear_timing(hook):
    """Profiling: ``hook(algorithmic_bytes) -> (start_event, stop_event) | None`` is asked for a torch event pair
    before every skinny-GEMM launch (recorded by the C ABI around the kernel, on the launch stream)."""
    global _linear_timing
    _linear_timing = hook


class NormFold:
    """An RMSNorm that has not been applied yet: ``h`` is the un-normalised residual stream, ``ssq`` [rows, hidden / 64]
    the fp32 partial sums of squares of its rows (written by the projection that produced ``h``, ``linear(...,
    ssq_out=True)``), ``weight`` / ``eps`` the norm's parameters.  The projections take it as ``norm=`` together with
    ``x = h`` and normalise on the way in -- bit-identical to ``rmsnorm`` followed by the plain call."""
    __slots__ = ("weight", "eps", "ssq")

    def __init__(self, weight, eps, ssq):
        self.weight, self.eps, self.ssq = weight, float(eps), ssq


PREFETCH_PROBE = 0       # measurement switch (tools/bench_l2_prefetch.py), 0 in the product


def _linear_call(x, weights, biases, epilogue, n_splits=0, timing=None, rope=None, residual=None, norm=None, ssq_out=False):
    _dev(x)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1 or x2.stride(0) % 8 != 0 or x2.data_ptr() % 16 != 0:
        x2 = x2.contiguous()
    M = x2.shape[0]
    d = _C.LinearDesc()
    d.x = x2.data_ptr()
    n_tot = 0
    for i, w in enumerate(weights):
        if not isinstance(w, PackedWeight):
            raise TypeError("linear: weights must be PackedWeight (ops.pack_weight(nn.Linear.weight))")
        if w.k != K or w.dtype != x.dtype:
            raise ValueError(f"packed weight [{w.n}, {w.k}] {w.dtype} does not match x [..., {K}] {x.dtype}")
        if w.rope != (epilogue == _C.LS_EPI_QKV_ROPE and i < 2):
            raise ValueError("linear: q/k weights of linear_qkv_rope are packed with rope=True, every other weight without")
        d.w[i] = w.data.data_ptr()
        b = biases[i] if biases is not None else None
        if b is not None and (b.dtype != x.dtype or not b.is_contiguous() or b.numel() != w.n):
            raise ValueError("bias must be a contiguous [N] tensor of x's dtype")
        d.bias[i] = b.data_ptr() if b is not None else None
        d.n[i] = w.n
        n_tot += w.n
    d.n_seg = len(weights)
    n_out = weights[0].n if epilogue == _C.LS_EPI_SILU_MUL else n_tot
    y = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    d.y = y.data_ptr()
    d.M, d.K = M, K
    d.dtype = _dtype(x)
    d.epilogue = epilogue
    d.n_splits = n_splits
    d.ldx, d.ldy = x2.stride(0), n_out
    if residual is not None:
        _dev(residual)
        r2 = residual.reshape(-1, residual.shape[-1])
        if r2.shape != (M, n_out) or r2.dtype != x.dtype or r2.stride(-1) != 1 or r2.stride(0) % 4 != 0:
            raise ValueError("linear: residual must be [..., N] of x's dtype with contiguous rows")
        d.residual, d.ldr = r2.data_ptr(), r2.stride(0)
    if rope is not None:
        cos, sin = rope
        _dev(cos, sin)
        for t in (cos, sin):
            if t.dtype != x.dtype or not t.is_contiguous() or t.numel() != M * 128:
                raise ValueError("linear_qkv_rope: cos/sin must be contiguous [rows, 128] tables of x's dtype")
        d.rope_cos, d.rope_sin = cos.data_ptr(), sin.data_ptr()
    ssq = None
    if norm is not None:
        _dev(norm.weight, norm.ssq)
        if (norm.weight.dtype != x.dtype or norm.weight.numel() != K or not norm.weight.is_contiguous()
                or norm.ssq.dtype != torch.float32 or norm.ssq.dim() != 2 or norm.ssq.shape[0] != M or not norm.ssq.is_contiguous()):
            raise ValueError("linear: norm = NormFold(weight [K] of x's dtype, eps, ssq fp32 [rows, parts] contiguous)")
        d.norm_weight, d.norm_eps = norm.weight.data_ptr(), norm.eps
        d.ssq_in, d.ssq_parts = norm.ssq.data_ptr(), norm.ssq.shape[1]
    if ssq_out:
        if epilogue != _C.LS_EPI_NONE or len(weights) != 1 or n_out % 64:
            raise ValueError("linear: ssq_out goes with a single plain projection whose N is a multiple of 64")
        ssq = torch.empty((M, n_out // 64), dtype=torch.float32, device=x.device)
        d.ssq_out = ssq.data_ptr()
    if timing is None and _linear_timing is not None:
        rows_w = sum(w.n for w in weights) * (2 if epilogue == _C.LS_EPI_SILU_MUL else 1)
        timing = _linear_timing((rows_w * K + M * K + M * n_out) * x.element_size())
    if timing is not None:          # (torch.cuda.Event, torch.cuda.Event), both already created by a record()
        d.ev_start, d.ev_stop = timing[0].cuda_event, timing[1].cuda_event
    lib = _C.load()
    key = (M, K, d.n[0], d.n[1], d.n[2], epilogue, n_splits, d.dtype, norm is not None)
    need = _linear_need.get(key)
    if need is None:
        need = lib.ls_linear_workspace_bytes(C.byref(d))
        if need == 0:
            _C.check(lib.ls_linear_fwd(C.byref(d), None, 0, _stream()), "ls_linear_fwd")      # raises with the reason
        _linear_need[key] = need
    stream = _stream()
    ws = _gemm_ws.get(x.device, need, stream)
    if PREFETCH_PROBE > 0:          # tools/bench_l2_prefetch.py: what an L2-resident head of the weight stream is worth
        ev = (d.ev_start, d.ev_stop)
        d.ev_start, d.ev_stop = None, None
        _C.check(lib.ls_linear_prefetch(C.byref(d), PREFETCH_PROBE, ws.data_ptr(), ws.numel(), stream), "ls_linear_prefetch")
        d.ev_start, d.ev_stop = ev
    _C.check(lib.ls_linear_fwd(C.byref(d), ws.data_ptr(), ws.numel(), stream), "ls_linear_fwd")
    return (y, ssq) if ssq_out else y


def linear(x: torch.Tensor, weight: PackedWeight, bias: Optional[torch.Tensor] = None, n_splits: int = 0, timing=None,
           residual: Optional[torch.Tensor] = None, norm: Optional[NormFold] = None, ssq_out: bool = False):
    """``F.linear(x, weight, bias)`` for M <= 80 token rows: ``[..., K] -> [..., N]``; with ``residual`` [..., N]:
    ``residual + F.linear(...)`` (the projection rounded first, as the two separate operators do).
    ``norm``: ``F.linear(rmsnorm(x), ...)`` (see NormFold).  ``ssq_out``: returns ``(y, ssq)`` with the partial sums of
    squares of y's rows, the input of the NormFold of the projection that consumes y."""
    out = _linear_call(x, [weight], [bias], _C.LS_EPI_NONE, n_splits, timing, residual=residual, norm=norm, ssq_out=ssq_out)
    if ssq_out:
        return out[0].view(*x.shape[:-1], weight.n), out[1]
    return out.view(*x.shape[:-1], weight.n)


def linear_multi(x: torch.Tensor, weights, biases=None, n_splits: int = 0, timing=None, norm: Optional[NormFold] = None):
    """Several linears of the same input in ONE launch (q|k|v): returns views ``[..., n_i]`` of one
    ``[M, sum n_i]`` buffer."""
    weights = list(weights)
    biases = list(biases) if biases is not None else [None] * len(weights)
    y = _linear_call(x, weights, biases, _C.LS_EPI_NONE, n_splits, timing, norm=norm)
    outs, o = [], 0
    for w in weights:
        outs.append(y[:, o:o + w.n].unflatten(0, x.shape[:-1]))
        o += w.n
    return outs


def linear_qkv_rope(x: torch.Tensor, weights, biases, cos: torch.Tensor, sin: torch.Tensor, n_splits: int = 0, timing=None,
                    norm: Optional[NormFold] = None):
    """``apply_rotary_pos_emb(q_proj(x), k_proj(x), cos, sin)`` and ``v_proj(x)`` (``llama.py:371-378``) in ONE launch:
    the rotation runs in the projection's epilogue on the rounded outputs, with the roundings of ``rope_apply_``.
    ``weights`` = [q] or [q, k] or [q, k, v] (q, k packed with ``rope=True``); cos/sin [rows, 128]."""
    weights = list(weights)
    biases = list(biases) if biases is not None else [None] * len(weights)
    y = _linear_call(x, weights, biases, _C.LS_EPI_QKV_ROPE, n_splits, timing, rope=(cos, sin), norm=norm)
    outs, o = [], 0
    for w in weights:
        outs.append(y[:, o:o + w.n].unflatten(0, x.shape[:-1]))
        o += w.n
    return outs


def mlp_gate_up(x: torch.Tensor, gate_up: PackedWeight, n_splits: int = 0, timing=None, norm: Optional[NormFold] = None):
    """``act_fn(gate_proj(x)) * up_proj(x)`` (SiLU) with the reference's roundings: both projections and the
    activation are rounded to the storage dtype before the product (qwen2.py:229).  ``gate_up`` = pack_gate_up(...)."""
    y = _linear_call(x, [gate_up], None, _C.LS_EPI_SILU_MUL, n_splits, timing, norm=norm)
    return y.view(*x.shape[:-1], gate_up.n)


def causal_mask_bits(n: int, device) -> torch.Tensor:
    """Packed lower-triangular mask [1, n, words]: row r sees new keys j <= r."""
    key = (n, device.index)
    t = _causal_bits.get(key)
    if t is None:
        words = (n + 31) // 32
        rows = []
        for r in range(n):
            v = (1 << (r + 1)) - 1
            rows.append([(v >> (32 * w)) & 0xFFFFFFFF for w in range(words)])
        t = torch.tensor(rows, dtype=torch.int64).to(torch.int32).view(1, n, words).to(device)   # wraps to the same bits
        _causal_bits[key] = t
    return t


def pack_tree_mask(tree_mask: torch.Tensor) -> torch.Tensor:
    """int64 [b,M,N] 0/1 mask -> packed uint32 [b,M,ceil(N/32)] (stored as int32)."""
    _dev(tree_mask)
    if tree_mask.dtype != torch.int64:
        tree_mask = tree_mask.to(torch.int64)
    tree_mask = tree_mask.contiguous()
    b, M, N = tree_mask.shape
    words = (N + 31) // 32
    bits = torch.empty((b, M, words), dtype=torch.int32, device=tree_mask.device)
    lib = _C.load()
    _C.check(lib.ls_pack_tree_mask(tree_mask.data_ptr(), b, M, N, bits.data_ptr(), words, _stream()), "ls_pack_tree_mask")
    return bits


def _check_qkv(q, k_cache, v_cache):
    if q.dim() != 4 or q.shape[-1] != 128:
        raise ValueError(f"q must be [b,sq,H,128], got {tuple(q.shape)}")
    if k_cache.dim() != 4 or k_cache.shape[-1] != 128 or k_cache.shape != v_cache.shape:
        raise ValueError("k_cache/v_cache must be [b,S,Hkv,128] and equal-shaped")
    if q.stride(-1) != 1 or k_cache.stride(-1) != 1 or v_cache.stride(-1) != 1:
        raise ValueError("innermost dimension must be contiguous")
    if k_cache.stride() != v_cache.stride():
        raise ValueError("k_cache and v_cache must share strides")
    if k_cache.dtype != q.dtype or v_cache.dtype != q.dtype:
        raise TypeError("q/k_cache/v_cache dtypes differ")


def _desc(q, k_cache, v_cache, cache_seqlens, kv_len_hint, k_new=None, v_new=None, mask_bits=None, out=None,
          lse=None, new_mode=LS_NEW_NONE, n_new=0, n_new_cached=0, scatter_new=0, causal=False, window_left=-1,
          n_app=0, prescale_q=False, softmax_scale=None, n_splits=0, timing=None) -> AttnDesc:
    b, sq, H, D = q.shape
    Hkv = k_cache.shape[2]
    d = AttnDesc()
    d.q = q.data_ptr()
    d.k_cache = k_cache.data_ptr()
    d.v_cache = v_cache.data_ptr()
    d.k_new = k_new.data_ptr() if k_new is not None else None
    d.v_new = v_new.data_ptr() if v_new is not None else None
    if cache_seqlens.dtype != torch.int32:
        cache_seqlens = cache_seqlens.to(torch.int32)
    d._keep = (cache_seqlens,)          # keep the converted tensor alive for the launch
    d.cache_seqlens = cache_seqlens.data_ptr()
    d.mask_bits = mask_bits.data_ptr() if mask_bits is not None else None
    d.out = out.data_ptr() if out is not None else None
    d.lse = lse.data_ptr() if lse is not None else None
    if timing is not None:          # (torch.cuda.Event, torch.cuda.Event), both already created by a record()
        d.ev_start, d.ev_stop = timing[0].cuda_event, timing[1].cuda_event
    d.b, d.sq, d.H, d.Hkv = b, sq, H, Hkv
    d.dtype = _dtype(q)
    d.new_mode = new_mode
    d.n_new = n_new
    d.n_new_cached = n_new_cached
    d.mask_words = mask_bits.shape[-1] if mask_bits is not None else 0
    d.scatter_new = int(scatter_new)
    d.causal = int(causal)
    d.window_left = int(window_left)
    d.n_app = int(n_app)
    d.prescale_q = int(prescale_q)
    d.kv_len_hint = int(kv_len_hint)
    d.n_splits = int(n_splits)
    d.softmax_scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    d.q_stride_b, d.q_stride_s, d.q_stride_h = q.stride(0), q.stride(1), q.stride(2)
    d.kc_stride_b, d.kc_stride_s, d.kc_stride_h = k_cache.stride(0), k_cache.stride(1), k_cache.stride(2)
    if k_new is not None:
        if k_new.stride() != v_new.stride() or k_new.stride(-1) != 1:
            raise ValueError("k_new/v_new must share strides and be contiguous in the last dim")
        d.kn_stride_b, d.kn_stride_s, d.kn_stride_h = k_new.stride(0), k_new.stride(1), k_new.stride(2)
    if out is not None:
        d.out_stride_b, d.out_stride_s, d.out_stride_h = out.stride(0), out.stride(1), out.stride(2)
    return d


def _hint(cache_seqlens, kv_len_hint, k_cache):
    if kv_len_hint is None:
        # no host-side bound known: size the grid for the whole cache allocation (no sync)
        return int(k_cache.shape[1])
    return int(kv_len_hint)


def _run(d: AttnDesc, device):
    lib = _C.load()
    nbytes = lib.ls_attn_workspace_bytes(C.byref(d))
    if nbytes == 0:
        _C.check(-1, "ls_attn_workspace_bytes")
    ws = _ws.get(device, nbytes)
    _C.check(lib.ls_attn_fwd(C.byref(d), ws.data_ptr(), ws.numel(), _stream()), "ls_attn_fwd")


def attn_kernel_name(M: int, verify: bool = True) -> str:
    """Name of the stage-1 kernel a verification call with ``M = (H / Hkv) * sq`` rows per kv head is served by
    (benchmark labels; ``ls_attn_kernel_name``)."""
    d = AttnDesc()
    d.b, d.sq, d.H, d.Hkv, d.dtype = 1, M, 1, 1, LS_F16
    d.window_left = -1
    d.q = d.k_cache = d.v_cache = d.cache_seqlens = 1            # non-null placeholders: nothing is dereferenced
    d.q_stride_s = d.q_stride_h = d.kc_stride_s = d.kc_stride_h = 128
    if verify:
        d.new_mode, d.n_new, d.mask_words, d.mask_bits, d.k_new, d.v_new = LS_NEW_TARGET, min(M, 74), 3, 1, 1, 1
    name = _C.load().ls_attn_kernel_name(C.byref(d))
    return name.decode() if name else "?"


def kvcache_attention(q, k_cache, v_cache, k=None, v=None, cache_seqlens=None, causal=False, window_size=(-1, -1),
                      return_softmax_lse=False, softmax_scale=None, kv_len_hint: Optional[int] = None, n_splits=0):
    """``flash_attn_with_kvcache`` as the reference uses it.  New
    ``k``/``v`` rows are appended IN PLACE at ``cache_seqlens`` and attended causally
    (the reference only appends with ``causal=True``)."""
    _dev(q, k_cache, v_cache, k, v, cache_seqlens)
    _check_qkv(q, k_cache, v_cache)
    b, sq, H, D = q.shape
    if window_size[1] not in (-1, 0):
        raise NotImplementedError("right window other than -1/0")
    out = torch.empty((b, sq, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((b, H, sq), dtype=torch.float32, device=q.device) if return_softmax_lse else None
    hint = _hint(cache_seqlens, kv_len_hint, k_cache)
    if k is not None:
        if not causal:
            raise NotImplementedError("append without causal=True is not a reference call pattern")
        n = k.shape[1]
        if n != sq:
            raise ValueError("appended rows must equal query rows")
        bits = causal_mask_bits(n, q.device).expand(b, -1, -1).contiguous()
        d = _desc(q, k_cache, v_cache, cache_seqlens, hint, k_new=k, v_new=v, mask_bits=bits, out=out, lse=lse,
                  new_mode=LS_NEW_FLASH, n_new=n, scatter_new=1, causal=True, window_left=window_size[0], n_app=n,
                  softmax_scale=softmax_scale, n_splits=n_splits)
    else:
        d = _desc(q, k_cache, v_cache, cache_seqlens, hint, out=out, lse=lse, causal=causal or window_size[1] == 0,
                  window_left=window_size[0], softmax_scale=softmax_scale, n_splits=n_splits)
    _run(d, q.device)
    return (out, lse) if return_softmax_lse else out


PREFILL_CHUNK = 64


def prefill_attention(q, k, v, k_cache, v_cache, window_left: int = -1, start: int = 0, total: Optional[int] = None):
    """Prompt attention (K13) through the decode kernels: the prompt is appended to the (empty)
    caches ``PREFILL_CHUNK`` rows at a time with the causal(+window) append form of
    ``kvcache_attention`` -- the same function as ``flash_attn_func(causal=True[, window])``
    (``llama.py:218``, ``llama_glide.py:227``), evaluated block-wise.  [b,L,H,128] -> [b,L,H,128];
    fills ``k_cache/v_cache[:, start:start + L]``.  ``start`` > 0: the rows are positions [start, start + L) of a
    prompt of ``total`` rows whose first ``start`` rows are already in the caches (sequence-sharded prefill).  The chunk
    size follows from the WHOLE prompt and chunk boundaries sit at multiples of it counted from global row 0, so a rank
    of a sharded prefill makes the very calls the single-GPU prefill makes for its rows -- except for the one chunk a shard
    boundary cuts in two (<= chunk rows per boundary: same function, other block split, equal within the fp16 tolerance)."""
    b, L, H, D = q.shape
    total = start + L if total is None else int(total)
    CH = prefill_chunk(total, H // k.shape[2], window_left)
    head = min(L, (-start) % CH)                 # rows up to the next global chunk boundary
    outs = []
    if head:
        lens = torch.full((b,), start, dtype=torch.int32, device=q.device)
        outs.append(kvcache_attention(q[:, :head], k_cache, v_cache, k[:, :head], v[:, :head], cache_seqlens=lens,
                                      causal=True, window_size=(window_left, -1), kv_len_hint=start))
    if b == 1 and (L - head) // CH >= 2:
        outs.append(_prefill_attention_batched(q[:, head:], k[:, head:], v[:, head:], k_cache, v_cache, window_left, start + head, CH))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
    for s0 in range(head, L, CH):
        s1 = min(L, s0 + CH)
        lens = torch.full((b,), start + s0, dtype=torch.int32, device=q.device)
        outs.append(kvcache_attention(q[:, s0:s1], k_cache, v_cache, k[:, s0:s1], v[:, s0:s1], cache_seqlens=lens,
                                      causal=True, window_size=(window_left, -1), kv_len_hint=start + s0))
    return torch.cat(outs, dim=1)


def prefill_chunk(total: int, g: int, window_left: int) -> int:
    """Prompt rows per chunk, a function of the WHOLE prompt (``total`` rows) only.  Long prompts of GQA models (g = H / Hkv
    >= 4) use g * chunk = 257..320 rows, which puts the prefix part of every chunk on the streaming MFMA kernel (ws_eligible:
    append chunks with kv_len_hint >= 4096); short prompts, windows and small groups keep 64 (the chunking the
    reference-generated goldens were checked against)."""
    if window_left < 0 and g >= 4 and total >= 8192 and 320 // g >= 32:
        return 320 // g
    return PREFILL_CHUNK


PREFILL_GROUP = 512       # prompt chunks per launch (bounds the fp32 workspace: 3 x group x 64 rows x H x 512 B)
PREFILL_SPLITS = 2        # key splits per chunk: with ONE, the ~32 chunks resident at a time walk the same K/V lines in lock
                          # step (measured 0.34 s per layer at 128k against 0.18 s); an ODD count leaves two of the eight XCDs
                          # with the light new-block workgroups only (grid.x = splits + 1, XCD = linear id % 8)


def _prefill_attention_batched(q, k, v, k_cache, v_cache, window_left, start, CH=PREFILL_CHUNK):
    """The same chunk-wise evaluation with the chunks of a prompt as the BATCH of one call: chunk c is batch element c
    with ``cache_seqlens[c] = start + 64 c`` over the one shared cache (batch stride 0; the prompt's K/V rows are copied
    into it first, so nothing is scattered by the launch), its 64 rows are the appended block with the causal mask.  One
    two key splits per element (thousands of workgroups as it is): no per-chunk launches, 16x fewer split partials -- a
    128k prompt is 4 launches per layer instead of 4096.  One layer (tools/bench_prefill.py): 16k 0.0046 s (0.0092 chunk
    by chunk), 64k 0.049 (0.060), 128k 0.181 (0.191)."""
    _dev(q, k, v, k_cache, v_cache)
    _check_qkv(q, k_cache, v_cache)
    b, L, H, D = q.shape
    Hkv = k.shape[2]
    n_full = L // CH
    k_cache[:, start:start + L] = k
    v_cache[:, start:start + L] = v
    out = torch.empty((b, L, H, D), dtype=q.dtype, device=q.device)
    bits1 = causal_mask_bits(CH, q.device)
    kc = k_cache.as_strided((1, k_cache.shape[1], Hkv, D), (0, k_cache.stride(1), k_cache.stride(2), 1))
    vc = v_cache.as_strided((1, v_cache.shape[1], Hkv, D), (0, v_cache.stride(1), v_cache.stride(2), 1))
    for c0 in range(0, n_full, PREFILL_GROUP):
        n = min(PREFILL_GROUP, n_full - c0)
        r0 = c0 * CH

        def chunks(t):      # [1, L, h, D] rows r0 .. r0 + n*CH as [n, CH, h, D] without a copy
            return t.as_strided((n, CH, t.shape[2], D), (CH * t.stride(1), t.stride(1), t.stride(2), 1),
                                t.storage_offset() + r0 * t.stride(1))
        lens = (torch.arange(n, dtype=torch.int32, device=q.device) * CH + (start + r0))
        d = _desc(chunks(q), kc.expand(n, -1, -1, -1), vc.expand(n, -1, -1, -1), lens, start + r0 + n * CH, k_new=chunks(k),
                  v_new=chunks(v), mask_bits=bits1.expand(n, -1, -1).contiguous(), out=chunks(out), new_mode=LS_NEW_FLASH,
                  n_new=CH, scatter_new=0, causal=True, window_left=window_left, n_app=CH, n_splits=PREFILL_SPLITS)
        _run(d, q.device)
    if n_full * CH < L:       # the ragged tail: one ordinary call
        s0 = n_full * CH
        lens = torch.full((b,), start + s0, dtype=torch.int32, device=q.device)
        out[:, s0:] = kvcache_attention(q[:, s0:], k_cache, v_cache, k[:, s0:], v[:, s0:], cache_seqlens=lens, causal=True,
                                        window_size=(window_left, -1), kv_len_hint=start + s0)
    return out


def verify_attention(q, k_new, v_new, k_cache, v_cache, cache_lens, mask_bits, last_layer: bool,
                     softmax_scale: float = 1.0 / (128 ** 0.5), kv_len_hint: Optional[int] = None, n_splits=0, timing=None):
    """Hybrid tree-verification attention of one target layer (K1+K2+K3 fused):
    prefix flash-decoding over ``cache[:, :cache_lens]`` + scatter of the R new K/V rows at
    ``cache_lens`` + tree-masked part + fp16 merge.  ``mask_bits`` = ``pack_tree_mask`` of
    the [b,R,R] verification mask.  Returns [b,R,H,128]."""
    _dev(q, k_new, v_new, k_cache, v_cache, cache_lens, mask_bits)
    _check_qkv(q, k_cache, v_cache)
    b, R, H, D = q.shape
    out = torch.empty((b, R, H, D), dtype=q.dtype, device=q.device)
    d = _desc(q, k_cache, v_cache, cache_lens, _hint(cache_lens, kv_len_hint, k_cache), k_new=k_new, v_new=v_new,
              mask_bits=mask_bits, out=out, new_mode=LS_NEW_TARGET, n_new=R, scatter_new=1, prescale_q=last_layer,
              softmax_scale=softmax_scale, n_splits=n_splits, timing=timing)
    _run(d, q.device)
    return out


def draft_tree_attention(q, k_new, v_new, k_cache, v_cache, cache_lens, mask_bits, N: int, window: int = 512,
                         kv_len_hint: Optional[int] = None):
    """Draft self-attention of a tree step (K5+K6 fused): window prefix over rows [0,p) +
    scatter of the M new rows at ``p + [N-M, N)`` + tree-masked part over rows ``p + [0,N)``
    + fp32 merge.  ``cache_lens`` = p.  Returns [b,M,H,128]."""
    _dev(q, k_new, v_new, k_cache, v_cache, cache_lens, mask_bits)
    _check_qkv(q, k_cache, v_cache)
    b, M, H, D = q.shape
    out = torch.empty((b, M, H, D), dtype=q.dtype, device=q.device)
    d = _desc(q, k_cache, v_cache, cache_lens, _hint(cache_lens, kv_len_hint, k_cache), k_new=k_new, v_new=v_new,
              mask_bits=mask_bits, out=out, new_mode=LS_NEW_DRAFT, n_new=N, n_new_cached=N - M, scatter_new=1,
              window_left=window)
    _run(d, q.device)
    return out


def tree_attention(q, k, v, tree_mask, sm_scale=None):
    """``triton_tree_attn.attention(q, k, v, tree_mask)`` (``longspec/test/triton_tree_attn.py:19-77``):
    q [B,H,M,128], k/v [B,Hkv,N,128], tree_mask [B,M,N] (non-zero = visible) -> (o [B,H,M,128], L [B,H,M] fp32).
    Runs the draft-mode new-block path over an empty prefix."""
    _dev(q, k, v, tree_mask)
    B, H, M, D = q.shape
    Hkv, N = k.shape[1], k.shape[2]
    qt = q.permute(0, 2, 1, 3).contiguous()
    kt = k.permute(0, 2, 1, 3).contiguous()
    vt = v.permute(0, 2, 1, 3).contiguous()
    out = torch.empty((B, M, H, D), dtype=q.dtype, device=q.device)
    L = torch.empty((B, H, M), dtype=torch.float32, device=q.device)
    zeros = torch.zeros((B,), dtype=torch.int32, device=q.device)
    bits = pack_tree_mask(tree_mask)
    # the keys double as an (unread) one-row-per-key "cache": cache_seqlens = 0 makes the prefix empty
    d = _desc(qt, kt, vt, zeros, 0, k_new=kt, v_new=vt, mask_bits=bits, out=out, lse=L, new_mode=LS_NEW_DRAFT, n_new=N,
              softmax_scale=sm_scale)
    _run(d, q.device)
    return out.permute(0, 2, 1, 3), L


def lse_merge(parts_o: torch.Tensor, parts_lse: torch.Tensor, dtype=None, want_o32=False):
    """parts_o [W,b,sq,H,128] fp32, parts_lse [W,b,H,sq] fp32 -> merged out (dtype) and/or (o32, lse)."""
    _dev(parts_o, parts_lse)
    W, b, sq, H, D = parts_o.shape
    parts_o = parts_o.contiguous()
    parts_lse = parts_lse.contiguous()
    out = torch.empty((b, sq, H, D), dtype=dtype, device=parts_o.device) if dtype is not None else None
    o32 = torch.empty((b, sq, H, D), dtype=torch.float32, device=parts_o.device) if want_o32 else None
    lse = torch.empty((b, H, sq), dtype=torch.float32, device=parts_o.device)
    lib = _C.load()
    _C.check(lib.ls_lse_merge(parts_o.data_ptr(), parts_lse.data_ptr(), W, b, sq, H,
                              _DT[dtype] if dtype is not None else 0,
                              out.data_ptr() if out is not None else None,
                              o32.data_ptr() if o32 is not None else None, lse.data_ptr(), _stream()), "ls_lse_merge")
    return out, o32, lse


# ---- multi-GPU building blocks (sequence-sharded prefix KV) -------------------------------------
class ShardedAttnCall:
    """Stage 1 / exchange / stage 2 of one attention call whose prefix KV is sharded by sequence
    over ranks.  ``partial(sendbuf)`` writes this rank's normalised prefix partial
    ``[o32 (b*sq*H*128) | lse (b*H*sq)]`` into ``sendbuf`` (fp32); after an all-gather of those
    records, ``finish(gathered)`` merges the W records in rank order (deterministic, identical on
    every rank) with the locally computed new-block part."""

    def __init__(self, desc: AttnDesc, out: torch.Tensor, device):
        self.d = desc
        self.out = out
        self.device = device
        lib = _C.load()
        nbytes = lib.ls_attn_workspace_bytes(C.byref(desc))
        if nbytes == 0:
            _C.check(-1, "ls_attn_workspace_bytes")
        self.ws = _ws.get(device, nbytes)       # stream-ordered: stage 2 runs before the next call reuses it
        self.n_o = desc.b * desc.sq * desc.H * 128
        self.n_lse = desc.b * desc.H * desc.sq
        self.xchg_timing = None                 # optional (start, stop) torch events around the exchange + merge (bench.py)

    @property
    def record_floats(self) -> int:
        return self.n_o + self.n_lse

    def partial(self, sendbuf: torch.Tensor) -> torch.Tensor:
        lib = _C.load()
        d = self.d
        assert sendbuf.dtype == torch.float32 and sendbuf.numel() >= self.record_floats and sendbuf.is_contiguous()
        _C.check(lib.ls_attn_partial(C.byref(d), self.ws.data_ptr(), self.ws.numel(), _stream()), "ls_attn_partial")
        _C.check(lib.ls_attn_reduce_local(C.byref(d), self.ws.data_ptr(), self.ws.numel(), sendbuf.data_ptr(),
                                          sendbuf.data_ptr() + 4 * self.n_o, _stream()), "ls_attn_reduce_local")
        return sendbuf

    def attend_peer(self, xchg) -> torch.Tensor:
        """partial -> reduce + peer stores -> wait + merge: the exchange rides inside the two combine kernels
        (``xchg`` = the ``ls_xchg*`` of ``dist.PeerExchange``).  Kernel launches only: graph-capturable."""
        lib = _C.load()
        d = self.d
        ws, n, st = self.ws.data_ptr(), self.ws.numel(), _stream()
        _C.check(lib.ls_attn_partial(C.byref(d), ws, n, st), "ls_attn_partial")
        if self.xchg_timing is not None:
            self.xchg_timing[0].record()
        _C.check(lib.ls_attn_reduce_push(C.byref(d), ws, n, xchg, st), "ls_attn_reduce_push")
        _C.check(lib.ls_attn_finish_xchg(C.byref(d), xchg, ws, n, st), "ls_attn_finish_xchg")
        if self.xchg_timing is not None:
            self.xchg_timing[1].record()
        return self.out

    def finish(self, gathered: torch.Tensor) -> torch.Tensor:
        """gathered: fp32 [W, record_floats] (row w = rank w's record)."""
        lib = _C.load()
        assert gathered.dtype == torch.float32 and gathered.is_contiguous() and gathered.shape[1] >= self.record_floats
        W, stride = gathered.shape[0], gathered.stride(0)
        _C.check(lib.ls_attn_finish(C.byref(self.d), gathered.data_ptr(), gathered.data_ptr() + 4 * self.n_o, W, stride,
                                    stride, self.ws.data_ptr(), self.ws.numel(), _stream()), "ls_attn_finish")
        return self.out


def sharded_verify_attention(q, k_new, v_new, k_cache, v_cache, local_lens, mask_bits, last_layer,
                             softmax_scale=1.0 / (128 ** 0.5), kv_len_hint=None, timing=None) -> ShardedAttnCall:
    """Like ``verify_attention`` but ``k_cache/v_cache`` hold only this rank's prefix rows
    (``local_lens`` valid rows); the new rows are scattered at ``local_lens`` of the LOCAL cache
    (only the tail-owning rank's copy is ever read back as prefix)."""
    _dev(q, k_new, v_new, k_cache, v_cache, local_lens, mask_bits)
    _check_qkv(q, k_cache, v_cache)
    b, R, H, D = q.shape
    out = torch.empty((b, R, H, D), dtype=q.dtype, device=q.device)
    d = _desc(q, k_cache, v_cache, local_lens, _hint(local_lens, kv_len_hint, k_cache), k_new=k_new, v_new=v_new,
              mask_bits=mask_bits, out=out, new_mode=LS_NEW_TARGET, n_new=R, scatter_new=1, prescale_q=last_layer,
              softmax_scale=softmax_scale, timing=timing)
    return ShardedAttnCall(d, out, q.device)


def sharded_prefix_attention(q, k_cache, v_cache, local_lens, causal=False, kv_len_hint=None) -> ShardedAttnCall:
    """Prefix attention over a local KV shard (draft cross-attention, K7).  ``causal`` only on the
    rank that owns the tail of the sequence (all keys of the other shards precede every query)."""
    _dev(q, k_cache, v_cache, local_lens)
    _check_qkv(q, k_cache, v_cache)
    b, R, H, D = q.shape
    out = torch.empty((b, R, H, D), dtype=q.dtype, device=q.device)
    d = _desc(q, k_cache, v_cache, local_lens, _hint(local_lens, kv_len_hint, k_cache), out=out, causal=causal)
    return ShardedAttnCall(d, out, q.device)


# ---- RMSNorm / RoPE ----------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None):
    """y = LlamaRMSNorm(x [+ residual]).  With ``residual`` returns (y, x + residual)."""
    _dev(x, weight, residual)
    hidden = x.shape[-1]
    xc = x.contiguous()
    rows = xc.numel() // hidden
    y = torch.empty_like(xc)
    lib = _C.load()
    if residual is not None:
        rc = residual.contiguous()
        s = torch.empty_like(xc)
        _C.check(lib.ls_rmsnorm_fwd(xc.data_ptr(), rc.data_ptr(), weight.data_ptr(), y.data_ptr(), s.data_ptr(), rows,
                                    hidden, eps, _dtype(x), _stream()), "ls_rmsnorm_fwd")
        return y.view(x.shape), s.view(x.shape)
    _C.check(lib.ls_rmsnorm_fwd(xc.data_ptr(), None, weight.data_ptr(), y.data_ptr(), None, rows, hidden, eps,
                                _dtype(x), _stream()), "ls_rmsnorm_fwd")
    return y.view(x.shape)


def rope_cos_sin(position_ids: torch.Tensor, inv_freq: torch.Tensor, attention_scaling: float, dtype):
    """position_ids [b,R] int64, inv_freq [64] fp32 -> cos, sin [b,R,128] dtype."""
    _dev(position_ids, inv_freq)
    pos = position_ids.to(torch.int64).contiguous()
    b, R = pos.shape
    cos = torch.empty((b, R, 128), dtype=dtype, device=pos.device)
    sin = torch.empty((b, R, 128), dtype=dtype, device=pos.device)
    lib = _C.load()
    _C.check(lib.ls_rope_cos_sin(pos.data_ptr(), inv_freq.data_ptr(), float(attention_scaling), cos.data_ptr(),
                                 sin.data_ptr(), b * R, _DT[dtype], _stream()), "ls_rope_cos_sin")
    return cos, sin


def rope_apply_(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor):
    """In-place apply_rotary_pos_emb(unsqueeze_dim=2) on q [b,R,Hq,128], k [b,R,Hk,128]
    (views into a fused QKV buffer are fine: only the row stride must be uniform)."""
    _dev(q, k, cos, sin)
    b, R, Hq, D = q.shape
    Hk = k.shape[2]
    for t in (q, k):
        if t.stride(-1) != 1 or t.stride(2) != 128 or (b > 1 and t.stride(0) != R * t.stride(1)):
            raise ValueError("rope_apply_: heads must be packed (stride 128) and rows uniformly strided")
    lib = _C.load()
    _C.check(lib.ls_rope_apply(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), b * R, Hq, Hk, q.stride(1),
                               k.stride(1), _dtype(q), _stream()), "ls_rope_apply")
    return q, k


def tree_positions(tree_mask: torch.Tensor, base: Optional[torch.Tensor]) -> torch.Tensor:
    """position_ids = tree_mask.sum(-1) - 1 + base[:, None]  (llama.py:575-577)."""
    _dev(tree_mask, base)
    tm = tree_mask.to(torch.int64).contiguous()
    b, M, N = tm.shape
    pos = torch.empty((b, M), dtype=torch.int64, device=tm.device)
    if base is not None and base.dtype != torch.int32:
        base = base.to(torch.int32)
    lib = _C.load()
    _C.check(lib.ls_tree_positions(tm.data_ptr(), base.data_ptr() if base is not None else None, b, M, N,
                                   pos.data_ptr(), _stream()), "ls_tree_positions")
    return pos


def _tree_state(tree_mask, all_spec, logp_sum=None):
    """The in-place tree operators work on the round's state tensors as they are."""
    b, Fn = all_spec.shape
    if tree_mask.dtype != torch.int64 or all_spec.dtype != torch.int64 or tuple(tree_mask.shape) != (b, Fn, Fn):
        raise TypeError("tree state: tree_mask [b,F,F] and all_spec [b,F] must be int64")
    if not (tree_mask.is_contiguous() and all_spec.is_contiguous()):
        raise ValueError("tree state tensors must be contiguous")
    if logp_sum is not None and (logp_sum.dtype != torch.float32 or tuple(logp_sum.shape) != (b, Fn) or not logp_sum.is_contiguous()):
        raise TypeError("tree state: logp_sum must be contiguous fp32 [b,F]")
    return b, Fn


def _len_i32(t, b, name):
    if t is None:
        return None
    if t.dtype != torch.int32 or t.numel() != b or not t.is_contiguous():
        raise TypeError(f"{name} must be a contiguous int32 [b] tensor (it is updated in place)")
    return t.data_ptr()


def tree_grow(tree_mask, all_spec, logp_sum, topk_vals, topk_idx, vocab: int, lo: int, mid: int,
              base: Optional[torch.Tensor] = None, base_add: int = 0, want_next: bool = True):
    """One more tree level, in place on the round's state (``llama_glide.py:1021-1027``, ``:1056-1075``):
    node mid+j = child of node ``lo + topk_idx[j] // vocab`` with token ``topk_idx[j] % vocab`` and cumulative
    log-prob ``topk_vals[j]``; its mask row = the father's row + the diagonal.  Returns (position_ids [b,k],
    packed mask bits [b,k,words]) of the new level for the draft pass that runs next (None, None when
    ``want_next`` is False); ``base`` [b] int32 is the draft cache length and is advanced by ``base_add`` first."""
    _dev(tree_mask, all_spec, logp_sum, topk_vals, topk_idx, base)
    b, Fn = _tree_state(tree_mask, all_spec, logp_sum)
    k = topk_idx.shape[-1]
    vals = topk_vals.to(torch.float32).contiguous().view(b, k)
    idx = topk_idx.to(torch.int64).contiguous().view(b, k)
    pos = bits = None
    words = (mid + k + 31) // 32
    if want_next:
        pos = torch.empty((b, k), dtype=torch.int64, device=idx.device)
        bits = torch.empty((b, k, words), dtype=torch.int32, device=idx.device)
    lib = _C.load()
    _C.check(lib.ls_tree_grow(tree_mask.data_ptr(), all_spec.data_ptr(), logp_sum.data_ptr(), vals.data_ptr(), idx.data_ptr(),
                              b, Fn, k, int(vocab), lo, mid, _len_i32(base, b, "base"), base_add,
                              pos.data_ptr() if want_next else None, bits.data_ptr() if want_next else None, words, _stream()),
             "ls_tree_grow")
    return pos, bits


def tree_verify_inputs(acc_ids, a: int, all_spec, tree_mask, cache_lens, R: int, bump: Optional[torch.Tensor] = None,
                       bump_add: int = 0):
    """Token ids, position ids and packed tree mask of the R-row verification pass (``llama_glide.py:1078-1086``):
    rows = [a accepted tokens | the F-1 tree nodes | pads].  ``bump`` [b] int32 += ``bump_add`` (:1076)."""
    _dev(acc_ids, all_spec, tree_mask, cache_lens, bump)
    b, Fn = _tree_state(tree_mask, all_spec)
    if acc_ids.dtype != torch.int64 or acc_ids.dim() != 2 or acc_ids.shape[1] < a or acc_ids.stride(1) != 1:
        raise TypeError("acc_ids must be int64 [b, >=a] with a contiguous last dimension")
    dev = all_spec.device
    words = (R + 31) // 32
    veri = torch.empty((b, R), dtype=torch.int64, device=dev)
    pos = torch.empty((b, R), dtype=torch.int64, device=dev)
    bits = torch.empty((b, R, words), dtype=torch.int32, device=dev)
    lib = _C.load()
    _C.check(lib.ls_tree_verify_inputs(acc_ids.data_ptr(), acc_ids.stride(0), a, all_spec.data_ptr(), tree_mask.data_ptr(), b, Fn,
                                       R, _len_i32(cache_lens, b, "cache_lens"), veri.data_ptr(), pos.data_ptr(),
                                       bits.data_ptr(), words, _len_i32(bump, b, "bump"), bump_add, _stream()),
             "ls_tree_verify_inputs")
    return veri, pos, bits


def tree_collapse(all_spec, all_llm_pred, tree_mask, cache_lens, non_leaf_len: int, max_acc: int,
                  k_cache: Optional[torch.Tensor] = None, v_cache: Optional[torch.Tensor] = None, cache_len_add: int = 0,
                  out_acc_ids: Optional[torch.Tensor] = None):
    """Accept/reject tree collapse + last-layer KV row move, one launch, no host sync.
    Returns (acc_ids [b,max_acc] int64 zero-padded, acc_num [b] int64, double_input [b] int32,
    index_mapping [b,max_acc] int64, -1 padded).  The moved rows start at ``cache_lens + cache_len_add``.
    ``out_acc_ids``: a persistent [b,max_acc] int64 buffer to write the accepted ids into."""
    _dev(all_spec, all_llm_pred, tree_mask, cache_lens, k_cache, v_cache)
    b, Fn = all_spec.shape
    dev = all_spec.device
    spec = all_spec.to(torch.int64).contiguous()
    pred = all_llm_pred.to(torch.int64).contiguous()
    tm = tree_mask.to(torch.int64).contiguous()
    cl = cache_lens.to(torch.int32).contiguous()
    if out_acc_ids is not None:
        if out_acc_ids.dtype != torch.int64 or tuple(out_acc_ids.shape) != (b, max_acc) or not out_acc_ids.is_contiguous():
            raise TypeError("tree_collapse: out_acc_ids must be a contiguous int64 [b, max_acc] tensor")
        acc_ids = out_acc_ids
    else:
        acc_ids = torch.empty((b, max_acc), dtype=torch.int64, device=dev)
    acc_num = torch.empty((b,), dtype=torch.int64, device=dev)
    dbl = torch.empty((b,), dtype=torch.int32, device=dev)
    imap = torch.empty((b, max_acc), dtype=torch.int64, device=dev)
    lib = _C.load()
    if k_cache is not None:
        if k_cache.stride() != v_cache.stride() or k_cache.stride(3) != 1 or k_cache.stride(2) != k_cache.shape[3]:
            raise ValueError("tree_collapse: cache rows must be contiguous [Hkv*D]")
        row_elems = k_cache.shape[2] * k_cache.shape[3]
        _C.check(lib.ls_tree_collapse(spec.data_ptr(), pred.data_ptr(), tm.data_ptr(), cl.data_ptr(), cache_len_add, b, Fn,
                                      non_leaf_len, max_acc, acc_ids.data_ptr(), acc_num.data_ptr(), dbl.data_ptr(),
                                      imap.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), k_cache.stride(0),
                                      k_cache.stride(1), row_elems, _dtype(k_cache), _stream()), "ls_tree_collapse")
    else:
        _C.check(lib.ls_tree_collapse(spec.data_ptr(), pred.data_ptr(), tm.data_ptr(), cl.data_ptr(), cache_len_add, b, Fn,
                                      non_leaf_len, max_acc, acc_ids.data_ptr(), acc_num.data_ptr(), dbl.data_ptr(),
                                      imap.data_ptr(), None, None, 0, 0, 0, 0, _stream()), "ls_tree_collapse")
    return acc_ids, acc_num, dbl, imap


MT_WORDS = 512           # raw Mersenne-Twister words handed to a stochastic verification (a walk uses <= ~3 per child drawn)


def tree_verify_stochastic(all_spec, tree_mask, llm_logits, spec_logp, temperature: float, mt_words, exp_noise, max_acc: int):
    """``LlamaGlide.verify_stochastic`` (``llama_glide.py:1177-1245``) on the device with pre-drawn randomness
    (``ls_tree_verify_stochastic``): returns (acc_ids [b,max_acc], acc_num [b], words_used [b] int32)."""
    _dev(all_spec, tree_mask, llm_logits, spec_logp, mt_words, exp_noise)
    b, Fn = all_spec.shape
    V = llm_logits.shape[-1]
    if llm_logits.stride(-1) != 1 or spec_logp.stride(-1) != 1 or spec_logp.dtype != torch.float32:
        raise ValueError("logits must be contiguous in the vocabulary, draft log-probs fp32")
    if mt_words.dtype != torch.int32 and mt_words.dtype != torch.uint32:
        raise TypeError("mt_words: 32-bit words")
    all_spec, tree_mask = all_spec.contiguous(), tree_mask.contiguous()
    exp_noise = exp_noise.to(llm_logits.dtype).contiguous()
    dev = all_spec.device
    acc_ids = torch.empty((b, max_acc), dtype=torch.int64, device=dev)
    acc_num = torch.empty((b,), dtype=torch.int64, device=dev)
    used = torch.empty((b,), dtype=torch.int32, device=dev)
    ws = torch.empty((b, V), dtype=torch.float32, device=dev)
    _C.check(_C.load().ls_tree_verify_stochastic(
        all_spec.data_ptr(), tree_mask.data_ptr(), llm_logits.data_ptr(), llm_logits.stride(0), llm_logits.stride(1),
        spec_logp.data_ptr(), spec_logp.stride(0), spec_logp.stride(1), b, Fn, spec_logp.shape[1], V, _dtype(llm_logits),
        float(temperature), mt_words.data_ptr(), mt_words.shape[-1], exp_noise.data_ptr(), acc_ids.data_ptr(), max_acc,
        acc_num.data_ptr(), used.data_ptr(), ws.data_ptr(), _stream()), "ls_tree_verify_stochastic")
    return acc_ids, acc_num, used


stochastic_noise_fn = None     # tests replay the reference's CPU generator: fn(V, dtype, device) -> Exponential(1) row


def verify_stochastic(input_ids, tree_mask, p_llm, p_ssm, temperature: float):
    """``LlamaGlide.verify_stochastic(input_ids, tree_mask, p_llm, p_ssm, temperature)`` (``llama_glide.py:1177-1245``).
    The randomness is the reference's: Python's ``random`` stream (choice / random) and one ``exponential_`` row per batch
    element from torch (what ``torch.multinomial(p, 1)`` draws); the walk itself runs on the device.  After the launch
    Python's generator is advanced by exactly the words the walk consumed.  One host read per batch row."""
    import random
    import numpy as np
    b, Fn = input_ids.shape
    V = p_llm.shape[-1]
    width = int(tree_mask.sum(-1).max()) + 1                      # :1193 (a host read in the reference too)
    acc_ids = input_ids.new_zeros((b, width))
    acc_num = input_ids.new_zeros((b,))
    for z in range(b):
        state = random.getstate()
        words = np.array([random.getrandbits(32) for _ in range(MT_WORDS)], dtype=np.uint32).view(np.int32)
        mt = torch.from_numpy(words).to(input_ids.device)[None]
        if stochastic_noise_fn is not None:
            noise = stochastic_noise_fn(V, p_llm.dtype, p_llm.device)
        else:
            noise = torch.empty((V,), dtype=p_llm.dtype, device=p_llm.device).exponential_(1)
        ids, num, used = tree_verify_stochastic(input_ids[z:z + 1], tree_mask[z:z + 1], p_llm[z:z + 1], p_ssm[z:z + 1],
                                                temperature, mt, noise[None], max(width, 2))
        used = int(used[0])
        if used < 0:
            raise RuntimeError("verify_stochastic: the pre-drawn random words were exhausted")
        random.setstate(state)
        for _ in range(used):
            random.getrandbits(32)
        acc_ids[z] = ids[0, :width]
        acc_num[z] = num[0]
    return acc_ids, acc_num


def tree_commit(acc_ids, acc_num, output_ids, emitted: int, eos: Optional[int], tree_mask, all_spec, logp_sum,
                target_lens: Optional[torch.Tensor] = None, target_add: int = 0,
                draft_kv_lens: Optional[torch.Tensor] = None, emitted_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """End of a round (``llama_glide.py:1093-1121``) in one launch: the accepted ids go to
    ``output_ids[:, emitted:]``, the tree state is reset for the next round (mask = root column, all_spec[0] =
    the last accepted id, log-prob sums = 0), ``target_lens += target_add``, ``draft_kv_lens += acc_num``.
    Returns state [b,2] int64 = (acc_num, whole-buffer EOS hit) -- the round's one host read.  ``emitted_dev`` [b]
    int32: the write offset lives on the device (read instead of ``emitted``, advanced by acc_num)."""
    _dev(acc_ids, acc_num, output_ids, tree_mask, all_spec, logp_sum, target_lens, draft_kv_lens, emitted_dev)
    b, Fn = _tree_state(tree_mask, all_spec, logp_sum)
    if output_ids.dtype != torch.int64 or output_ids.stride(1) != 1 or not acc_ids.is_contiguous() or acc_ids.dtype != torch.int64:
        raise TypeError("tree_commit: output_ids / acc_ids must be int64 with contiguous rows")
    state = torch.empty((b, 2), dtype=torch.int64, device=all_spec.device)
    lib = _C.load()
    _C.check(lib.ls_tree_commit(acc_ids.data_ptr(), acc_num.data_ptr(), b, acc_ids.shape[1], output_ids.data_ptr(),
                                output_ids.stride(0), output_ids.shape[1], emitted, _len_i32(emitted_dev, b, "emitted_dev"),
                                0 if eos is None else 1,
                                0 if eos is None else int(eos), state.data_ptr(), tree_mask.data_ptr(), all_spec.data_ptr(),
                                logp_sum.data_ptr(), Fn, _len_i32(target_lens, b, "target_lens"), target_add,
                                _len_i32(draft_kv_lens, b, "draft_kv_lens"), _stream()), "ls_tree_commit")
    return state


stochastic_uniform_fn = None   # tests replay the reference's CPU generator: fn(shape, device) -> U[0,1) fp32
stochastic_chain_noise_fn = None   # fn(shape, dtype, device) -> Exponential(1) of the model dtype


def chain_accept_stochastic(spec_logits, llm_verify_logits, spec_buffer, llm_verify_output):
    """The temperature > 0 branch of ``spec_generate`` (``llama_glide.py:715-736``): accept drafted token i with probability
    ``min(1, p_i / q_i)`` (q = soft-max of the draft's fp32 logits, p = soft-max of the target's logits in the model dtype,
    neither divided by the temperature -- as the reference), a rejected position takes one draw from p
    (``Categorical(p).sample()`` = ``argmax(p / p.sum / Exponential(1))``).  Rewrites ``llm_verify_output[:, :-1]`` in
    place and returns it with the accept mask [b, gamma] int64 for ``chain_commit``.  A handful of library kernels on
    gamma x V elements: this branch is not on the metric's path."""
    _dev(spec_logits, llm_verify_logits, spec_buffer, llm_verify_output)
    b, g1, V = llm_verify_logits.shape
    gamma = g1 - 1
    q_probs = torch.softmax(spec_logits[:, 1:, :], dim=-1)
    p_probs = torch.softmax(llm_verify_logits[:, :-1, :], dim=-1)
    idx = spec_buffer[:, 1:].unsqueeze(-1)
    alpha = torch.clip((torch.gather(p_probs, -1, idx).squeeze(-1) + 1e-9) / (torch.gather(q_probs, -1, idx).squeeze(-1) + 1e-9), 0.0, 1.0)
    u = stochastic_uniform_fn(alpha.shape, alpha.device) if stochastic_uniform_fn is not None else torch.rand_like(alpha)
    accept = u.lt(alpha)
    p2 = p_probs.reshape(-1, V)
    pn = p2 / p2.sum(-1, keepdim=True)
    nz = (stochastic_chain_noise_fn(pn.shape, pn.dtype, pn.device) if stochastic_chain_noise_fn is not None
          else torch.empty_like(pn).exponential_(1))
    resample = (pn / nz).argmax(dim=-1).reshape(b, gamma)
    llm_verify_output[:, :-1] = torch.where(accept, spec_buffer[:, 1:], resample)
    return llm_verify_output, accept.to(torch.int64).contiguous()


def chain_commit(llm_verify_output, spec_buffer, output_ids, cache_lens, draft_cache_lens, input_len, next_spec_start_token,
                 eos: Optional[int], accept_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """End of a chain-speculation round (``llama_glide.py:738-770``) in one launch: acceptance by cumulative match (or, at
    temperature > 0, by the cumulative product of ``accept_mask``, ``:732``), the verified ids + bonus token into
    ``output_ids``, the length bookkeeping and the next round's start tokens, in place.
    Returns state [b,2] int64 = (correct_len, EOS hit) -- the round's one host read."""
    _dev(llm_verify_output, spec_buffer, output_ids, cache_lens, draft_cache_lens, input_len, next_spec_start_token)
    b, g1 = llm_verify_output.shape
    for t, name in ((llm_verify_output, "llm_verify_output"), (spec_buffer, "spec_buffer"), (next_spec_start_token, "next_spec_start_token")):
        if t.dtype != torch.int64 or not t.is_contiguous():
            raise TypeError(f"chain_commit: {name} must be a contiguous int64 tensor")
    if tuple(spec_buffer.shape) != (b, g1) or tuple(next_spec_start_token.shape) != (b, 2):
        raise ValueError("chain_commit: spec_buffer [b, gamma+1], next_spec_start_token [b, 2]")
    if output_ids.dtype != torch.int64 or output_ids.stride(1) != 1:
        raise TypeError("chain_commit: output_ids must be int64 with contiguous rows")
    if accept_mask is not None and (accept_mask.dtype != torch.int64 or not accept_mask.is_contiguous() or tuple(accept_mask.shape) != (b, g1 - 1)):
        raise TypeError("chain_commit: accept_mask must be a contiguous int64 [b, gamma] tensor")
    state = torch.empty((b, 2), dtype=torch.int64, device=output_ids.device)
    lib = _C.load()
    _C.check(lib.ls_chain_commit(llm_verify_output.data_ptr(), spec_buffer.data_ptr(), b, g1 - 1, output_ids.data_ptr(),
                                 output_ids.stride(0), output_ids.shape[1], _len_i32(cache_lens, b, "cache_lens"),
                                 _len_i32(draft_cache_lens, b, "draft_cache_lens"), _len_i32(input_len, b, "input_len"),
                                 next_spec_start_token.data_ptr(), 0 if eos is None else 1, 0 if eos is None else int(eos),
                                 state.data_ptr(), accept_mask.data_ptr() if accept_mask is not None else None, _stream()),
             "ls_chain_commit")
    return state


EMBED_MAX_ROWS = 128
# one launch for embedding gather + RoPE table + first input norm of a decode pass (ls_pass_head); LONGSPEC_PASS_HEAD=0 keeps the
# three operators (A/B measurements; the results are bit-identical either way)
PASS_HEAD = os.environ.get("LONGSPEC_PASS_HEAD", "1") != "0"


def embed_supported(ids: torch.Tensor, weight: torch.Tensor) -> bool:
    """Short passes gather their embedding rows with ``ls_embed_rows``; prefill-sized inputs use the library."""
    return (ids.is_cuda and weight.is_cuda and 0 < ids.numel() <= EMBED_MAX_ROWS and weight.dtype in (torch.float16, torch.bfloat16)
            and weight.is_contiguous() and weight.shape[1] % 8 == 0)


def pass_head(weight: torch.Tensor, ids: torch.Tensor, inv_freq: torch.Tensor, attention_scaling: float, norm_weight: torch.Tensor,
              eps: float, position_ids: Optional[torch.Tensor] = None, pos_base: Optional[torch.Tensor] = None, pos_add: int = 0):
    """Head of a decode pass in one launch: ``embed_tokens(ids)``, the pass's RoPE table and the first layer's
    ``input_layernorm`` (``llama.py:579-580`` + ``LlamaDecoderLayer.forward``; ``llama_glide.py:1003-1006,1030-1033,437``).
    ids [b,q] int64; positions = ``position_ids`` [b,q] int64, or ``arange(q) + pos_base[:, None] + pos_add`` (pos_base [b] int32).
    Returns (embeds [b,q,hidden], normed [b,q,hidden], (cos, sin) [b,q,128]); bit-identical to the three separate operators."""
    _dev(weight, ids, inv_freq, norm_weight, position_ids, pos_base)
    b, q = ids.shape
    flat = ids.to(torch.int64).contiguous().view(-1)
    hidden = weight.shape[1]
    dev = weight.device
    buf = torch.empty((2, b, q, hidden), dtype=weight.dtype, device=dev)
    cs = torch.empty((2, b, q, 128), dtype=weight.dtype, device=dev)
    if position_ids is not None:
        pos = position_ids.to(torch.int64).contiguous()
        if pos.numel() != b * q:
            raise ValueError(f"pass_head: {pos.numel()} positions for {b * q} token rows")
        pos_p, base_p = pos.data_ptr(), None
    else:
        if pos_base is None or pos_base.numel() != b:
            raise ValueError("pass_head: position_ids [b,q] int64 or pos_base [b] int32")
        pos_base = pos_base if pos_base.dtype == torch.int32 else pos_base.to(torch.int32)
        pos_base = pos_base.contiguous()
        pos_p, base_p = None, pos_base.data_ptr()
    lib = _C.load()
    _C.check(lib.ls_pass_head(weight.data_ptr(), weight.shape[0], hidden, _dtype(weight), flat.data_ptr(), b * q, pos_p, base_p, q,
                              int(pos_add), inv_freq.data_ptr(), float(attention_scaling), norm_weight.data_ptr(), float(eps),
                              buf[0].data_ptr(), buf[1].data_ptr(), cs[0].data_ptr(), cs[1].data_ptr(), _stream()), "ls_pass_head")
    return buf[0], buf[1], (cs[0], cs[1])


def pass_head_supported(ids: torch.Tensor, weight: torch.Tensor, norm_weight: torch.Tensor) -> bool:
    return (embed_supported(ids, weight) and ids.dim() == 2 and ids.numel() <= 128 and weight.shape[1] // 8 <= 4096
            and norm_weight.dtype == weight.dtype and norm_weight.is_cuda)


def embed_rows(weight: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """``embed_tokens(ids)`` (``llama.py:579``): out[..., :] = weight[ids]."""
    _dev(weight, ids)
    flat = ids.to(torch.int64).contiguous().view(-1)
    out = torch.empty((flat.numel(), weight.shape[1]), dtype=weight.dtype, device=weight.device)
    lib = _C.load()
    _C.check(lib.ls_embed_rows(weight.data_ptr(), weight.shape[0], weight.shape[1], _dtype(weight), flat.data_ptr(), flat.numel(),
                               out.data_ptr(), _stream()), "ls_embed_rows")
    return out.view(*ids.shape, weight.shape[1])
