"""CPU restatement of the operators on LongSpec's draft-then-verify hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): imported by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg -- never by
``longspec_amd``.

Every function cites the reference lines it follows (paths relative to
``/root/reference/``).  Arithmetic is PyTorch-CPU: fp16 storage, fp32 compute,
with the reference's rounding points (casts to fp16) reproduced where the
reference has them ("reference-order" restatement).

Parity status
-------------
* ``flash_attn`` (third-party, ``flash_attn==2.6.3``, ``longspec/test/requirements.txt:2``)
  is NOT vendored in the reference and the reference has no tests at that seam:
  ``kvcache_attention`` / ``flash_attention`` restate the package's documented
  contract.  **Parity unpinned** for flash-attn's internal
  rounding/combine order; the *semantics* are pinned indirectly against the
  reference's own dense twins (``decoding_torch`` ``longspec/test/llama.py:161-197``,
  ``tree_decoding_torch`` ``longspec/train/models/llama.py:210-275``) and the
  lossless-generation property, via ``tests/golden``.
* Everything else (tree part, merge, Triton tree kernel, RMSNorm, RoPE,
  tree_verification, generation loops) is pinned against golden vectors produced
  by importing the reference itself in the build container
  (``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

F16 = torch.float16
F32 = torch.float32


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
def _mm_f16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """fp16 x fp16 matmul as the reference's GPU library does it: fp32
    accumulate, one rounding to the input dtype at the end."""
    dt = a.dtype
    return (a.float() @ b.float()).to(dt)


def _mul_scalar(x: torch.Tensor, s: float) -> torch.Tensor:
    """tensor(fp16) * python float, PyTorch semantics: fp32 op-math, scalar as
    fp32, one rounding to the tensor dtype."""
    return (x.float() * torch.tensor(s, dtype=F32)).to(x.dtype)


def _bottom_right_mask(sq: int, sk: int, causal: bool, window: Tuple[int, int]) -> Optional[torch.Tensor]:
    """flash-attn mask, bottom-right aligned: query row i
    may see key j iff  j <= i + sk - sq + right  and  j >= i + sk - sq - left.
    ``causal`` forces right = 0.  Returns a bool [sq, sk] "visible" mask or None."""
    left, right = window
    if causal:
        right = 0
    if left < 0 and right < 0:
        return None
    i = torch.arange(sq).view(-1, 1)
    j = torch.arange(sk).view(1, -1)
    vis = torch.ones(sq, sk, dtype=torch.bool)
    if right >= 0:
        vis &= j <= i + (sk - sq) + right
    if left >= 0:
        vis &= j >= i + (sk - sq) - left
    return vis


def _attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, vis: Optional[torch.Tensor],
            scale: float, keep_f32: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """One batch element.  q [sq,H,D], k/v [sk,Hkv,D] (fp16/bf16).  Flash-style
    numerics: fp32 scores, fp32 softmax statistics, P rounded to the input dtype
    before P.V, fp32 accumulation, one division by the *unrounded* row sum, one
    rounding of the output.  Returns (o [sq,H,D] in q.dtype, lse [H,sq] fp32)."""
    sq, H, D = q.shape
    sk, Hkv, _ = k.shape
    g = H // Hkv
    dt = q.dtype
    if sk == 0:       # empty key set: zeros and lse = -inf
        return torch.zeros_like(q, dtype=F32 if keep_f32 else dt), torch.full((H, sq), float("-inf"), dtype=F32)
    qf = q.float().permute(1, 0, 2)                                   # H sq D
    kf = k.float().permute(1, 0, 2).repeat_interleave(g, dim=0)       # H sk D
    vf = v.float().permute(1, 0, 2).repeat_interleave(g, dim=0)
    s = torch.matmul(qf, kf.transpose(1, 2)) * scale                  # H sq sk
    if vis is not None:
        s = s.masked_fill(~vis.unsqueeze(0), float("-inf"))
    m = s.max(dim=-1, keepdim=True).values
    m_safe = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = torch.exp(s - m_safe)
    l = p.sum(dim=-1, keepdim=True)
    o = torch.matmul(p.to(dt).float(), vf) / torch.where(l == 0, torch.ones_like(l), l)
    lse = (m_safe + torch.log(l)).squeeze(-1)                          # H sq  (-inf for empty rows)
    o = o.permute(1, 0, 2).contiguous()
    return (o if keep_f32 else o.to(dt)), lse


# --------------------------------------------------------------------------- #
# K1/K4/K5/K7/K13: the flash_attn contract (third-party; parity unpinned)
# --------------------------------------------------------------------------- #
def flash_attention(q, k, v, causal=False, window_size=(-1, -1), softmax_scale=None):
    """``flash_attn_func`` as used for prefill: ``longspec/test/llama.py:218``,
    ``longspec/test/llama_glide.py:227`` (window (512,-1), causal).
    q [b,sq,H,D], k/v [b,sk,Hkv,D] -> [b,sq,H,D]."""
    b, sq, H, D = q.shape
    sk = k.shape[1]
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    vis = _bottom_right_mask(sq, sk, causal, window_size)
    outs = [_attend(q[i], k[i], v[i], vis, scale)[0] for i in range(b)]
    return torch.stack(outs, 0)


def kvcache_attention(q, k_cache, v_cache, k=None, v=None, cache_seqlens=None, causal=False,
                      window_size=(-1, -1), return_softmax_lse=False, softmax_scale=None, keep_f32=False):
    """``flash_attn_with_kvcache`` as the reference calls it:
    target decode ``llama.py:324`` (append, causal); target verify prefix
    ``llama.py:385`` (no append, non-causal, LSE); draft self step 0
    ``llama_glide.py:261`` (append, causal, window 512); draft self tree prefix
    ``llama_glide.py:300`` (no append, NON-causal, window (512,-1), LSE; gotcha G3);
    draft cross ``llama_glide.py:265`` (causal) / ``:297`` (non-causal).

    New ``k``/``v`` rows are written IN PLACE into the caches at
    ``cache_seqlens[b] + [0, sk_new)``; rows >= the attended length are ignored.
    Returns out [b,sq,H,D] (and lse fp32 [b,H,sq])."""
    b, sq, H, D = q.shape
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    if cache_seqlens is None:
        cache_seqlens = torch.full((b,), k_cache.shape[1], dtype=torch.int32)
    if not torch.is_tensor(cache_seqlens):
        cache_seqlens = torch.full((b,), int(cache_seqlens), dtype=torch.int32)
    outs, lses = [], []
    for i in range(b):
        L = int(cache_seqlens[i])
        if k is not None:
            n_new = k.shape[1]
            k_cache[i, L:L + n_new] = k[i]
            v_cache[i, L:L + n_new] = v[i]
            sk = L + n_new
        else:
            sk = L
        vis = _bottom_right_mask(sq, sk, causal, window_size)
        o, lse = _attend(q[i], k_cache[i, :sk], v_cache[i, :sk], vis, scale, keep_f32=keep_f32)
        outs.append(o)
        lses.append(lse)
    out = torch.stack(outs, 0)
    if return_softmax_lse:
        return out, torch.stack(lses, 0)
    return out


# --------------------------------------------------------------------------- #
# a2/a3: target tree part + fp16 merge  (llama.py:385-421)
# --------------------------------------------------------------------------- #
def target_tree_part(q, k_new, v_new, tree_mask, prefix_lse, last_layer: bool,
                     softmax_scale: float = 1.0 / (128 ** 0.5)):
    """``LlamaAttention.tree_part_fwd`` (``longspec/test/llama.py:394-421``) minus
    the KV scatter (done by the caller, see ``target_verify_attention``).

    q [b,R,H,D], k_new/v_new [b,R,Hkv,D], tree_mask [b,R,R] (int, 0/1),
    prefix_lse [b,H,R] fp32.  Returns (current_out [b,R,H,D] fp16,
    weight [b,R,H,1] fp16).  Rounding points copied from the reference: fp16
    QK^T result, scale applied before (last layer) or after the matmul (G1),
    probabilities cast to fp16 before P.V (G2), fp16 P.V result, fp16 weight."""
    b, R, H, D = q.shape
    Hkv = k_new.shape[2]
    g = H // Hkv
    k = k_new.repeat_interleave(g, dim=2)
    v = v_new.repeat_interleave(g, dim=2)
    qh = q.transpose(1, 2)                 # b H R D
    kT = k.permute(0, 2, 3, 1)             # b H D R
    vh = v.transpose(1, 2)                 # b H R D
    if last_layer:
        score = _mm_f16(_mul_scalar(qh, softmax_scale), kT)          # llama.py:407
    else:
        score = _mul_scalar(_mm_f16(qh, kT), softmax_scale)          # llama.py:409
    score = score.to(F32)
    mask = tree_mask.unsqueeze(1).expand(-1, H, -1, -1)
    score = score.masked_fill(mask == 0, float("-inf"))              # llama.py:412
    attn_w = torch.softmax(score, dim=-1).to(q.dtype)                # llama.py:413
    current_out = _mm_f16(attn_w, vh).permute(0, 2, 1, 3)            # llama.py:414  b R H D
    current_lse = score.logsumexp(dim=-1, keepdim=True).transpose(1, 2)   # b R H 1
    p_lse = prefix_lse.reshape(b, H, R, -1).transpose(1, 2)          # llama.py:417-419
    weight = torch.sigmoid(p_lse - current_lse).to(q.dtype)          # llama.py:420
    return current_out, weight


def target_verify_attention(q, k_new, v_new, k_cache, v_cache, cache_lens, tree_mask,
                            last_layer: bool, softmax_scale: float = 1.0 / (128 ** 0.5)):
    """The hybrid tree-verification attention of one target layer, RoPE already
    applied: ``LlamaAttention.tree_decoding`` ``longspec/test/llama.py:385-387``
    + ``tree_part_fwd`` ``:394-421``.

    1. prefix: non-causal attention of all R rows over ``cache[:, :cache_lens]`` + LSE (``:385``)
    2. scatter k_new/v_new into the caches at ``cache_lens + [0,R)`` (``:396-399``) -- IN PLACE
    3. tree part (``target_tree_part``)
    4. fp16 merge ``prefix_o*w + current_out*(1-w)`` (``:387``)
    Returns attn_output [b,R,H,D] fp16 (before ``o_proj``)."""
    b, R, H, D = q.shape
    prefix_o, prefix_lse = kvcache_attention(q, k_cache, v_cache, cache_seqlens=cache_lens,
                                             return_softmax_lse=True, softmax_scale=softmax_scale)
    for i in range(b):
        L = int(cache_lens[i])
        k_cache[i, L:L + R] = k_new[i]
        v_cache[i, L:L + R] = v_new[i]
    current_out, weight = target_tree_part(q, k_new, v_new, tree_mask, prefix_lse, last_layer, softmax_scale)
    one = torch.ones((), dtype=q.dtype)
    return prefix_o * weight + current_out * (one - weight)           # llama.py:387 (all fp16)


def dense_tree_attention(q, k_new, v_new, k_cache, v_cache, cache_lens, tree_mask,
                         softmax_scale: float = 1.0 / (128 ** 0.5)):
    """The reference's own dense restatement of the hybrid path,
    ``tree_decoding_torch`` (``longspec/train/models/llama.py:248-266``): one
    softmax over [prefix | tree] keys.  b must be 1 (the reference indexes with
    ``cache_lens`` as a slice bound).  Used to show hybrid == dense up to fp16
    rounding.  Returns [b,R,H,D] fp16."""
    b, R, H, D = q.shape
    assert b == 1
    L = int(cache_lens[0])
    Hkv = k_new.shape[2]
    g = H // Hkv
    K = torch.cat((k_cache[:, :L], k_new), dim=1).repeat_interleave(g, dim=2)   # b L+R H D
    V = torch.cat((v_cache[:, :L], v_new), dim=1).repeat_interleave(g, dim=2)
    qh = q.transpose(1, 2)
    scores = _mul_scalar(_mm_f16(qh, K.permute(0, 2, 3, 1)), softmax_scale)      # :263
    total_mask = torch.cat((torch.zeros(b, R, L, dtype=torch.bool), tree_mask == 0), dim=2)
    scores = scores.masked_fill(total_mask.unsqueeze(1), float("-inf"))
    w = torch.softmax(scores.float(), dim=-1)
    out = _mm_f16(w.to(q.dtype), V.transpose(1, 2))
    return out.transpose(1, 2).contiguous()


# --------------------------------------------------------------------------- #
# a6: the Triton tree-attention kernel  (triton_tree_attn.py:115-251)
# --------------------------------------------------------------------------- #
def triton_tree_attention(q, k, v, tree_mask, sm_scale: Optional[float] = None, block_n: int = 32):
    """``triton_tree_attn.attention`` (``longspec/test/triton_tree_attn.py:19-77``)
    / ``_fwd_kernel`` (``:115-251``): q [B,H,M,D], k/v [B,Hkv,N,D] fp16,
    tree_mask [B,M,N] int (non-zero = visible).  Blocked online softmax over N in
    blocks of ``block_n`` = 32 (both ``get_fwd_config`` branches that apply at
    D=128, M<=1024 use BLOCK_N=32, ``:96,:111``), base-2 exponentials with
    ``qk_scale = sm_scale*log2(e)`` (``:141-142,218-219``), the extra causal mask
    ``P_SEQ + m >= n`` (``:213-214``), P cast to fp16 before P.V (``:227``),
    ``o = acc * (1/l)`` -> fp16 (``:242,248``), ``L = m*sm_scale + ln(l)`` (``:243``).
    Returns (o [B,H,M,D] fp16, L [B,H,M] fp32)."""
    B, H, M, D = q.shape
    Hk, N = k.shape[1], k.shape[2]
    g = H // Hk
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    P_SEQ = N - M
    log2e = 1.4426950408889634
    qk_scale = np.float32(sm_scale * log2e)
    o = torch.empty_like(q)
    Lout = torch.empty(B, H, M, dtype=F32)
    offs_m = torch.arange(M).view(-1, 1)
    for z in range(B):
        for h in range(H):
            hk = h // g
            qf = q[z, h].float()
            m_i = torch.full((M,), float("-inf"), dtype=F32)
            l_i = torch.zeros(M, dtype=F32)
            acc = torch.zeros(M, D, dtype=F32)
            for start in range(0, N, block_n):
                end = min(N, start + block_n)
                kb = k[z, hk, start:end].float()
                vb = v[z, hk, start:end]
                s = qf @ kb.t()
                tm = tree_mask[z, :, start:end]
                s = s + torch.where(tm != 0, torch.zeros((), dtype=F32), torch.full((), float("-inf"), dtype=F32))
                offs_n = torch.arange(start, end).view(1, -1)
                s = torch.where((P_SEQ + offs_m) >= offs_n, s, torch.full((), float("-inf"), dtype=F32))
                m_new = torch.maximum(m_i, s.max(dim=1).values)
                alpha = torch.exp2((m_i - m_new) * qk_scale)
                p = torch.exp2(s * qk_scale - (m_new * qk_scale).view(-1, 1))
                p_sum = p.sum(dim=1)
                acc = acc * alpha.view(-1, 1) + p.to(q.dtype).float() @ vb.float()
                l_i = l_i * alpha + p_sum
                m_i = m_new
            o[z, h] = (acc * (1.0 / l_i).view(-1, 1)).to(q.dtype)
            Lout[z, h] = m_i * np.float32(sm_scale) + torch.log(l_i)
    return o, Lout


# --------------------------------------------------------------------------- #
# a4/a5: draft (glide) attention
# --------------------------------------------------------------------------- #
def draft_self_attention_step0(q, k_new, v_new, k_cache, v_cache, cache_lens, window: int = 512):
    """Draft self-attention, step 0 of a round: ``GlideAttention.decoding``
    ``longspec/test/llama_glide.py:258-262`` -- append the ``a`` new rows at
    ``cache_lens``, causal, sliding window 512 (each row sees itself + 512
    previous keys).  In place on the caches.  Returns [b,a,H,D]."""
    return kvcache_attention(q, k_cache, v_cache, k_new, v_new, cache_seqlens=cache_lens,
                             causal=True, window_size=(window, -1))


def draft_cross_attention(q, k_llm, v_llm, llm_kv_len, causal: bool):
    """Draft cross-attention over the target's last-layer KV:
    ``llama_glide.py:264-265`` (step 0 / prefill, causal=True) and ``:297``
    (tree steps, causal=False); ``cache_seqlens = llm_kv_len``."""
    return kvcache_attention(q, k_llm, v_llm, cache_seqlens=llm_kv_len, causal=causal)


def draft_tree_self_attention(q, k_new, v_new, k_cache, v_cache, cache_lens, tree_mask, window: int = 512):
    """Draft self-attention of a tree step: ``GlideAttention.tree_decoding``
    ``longspec/test/llama_glide.py:300-302`` + ``triton_tree_part_fwd`` ``:309-329``.

    q [b,M,H,D]; tree_mask [b,M,N]; cache_lens = p (position of the tree root).
    1. prefix: NON-causal, window (512,-1) over rows [0,p) + LSE (``:300``; G3:
       row i sees keys j >= p - M + i - 512)
    2. scatter the M new K/V rows at ``p + [N-M, N)`` (``:312-315``), gather rows
       ``p + [0,N)`` (``:317-319``)
    3. Triton tree kernel -> (current_out fp16, L)
    4. fp32 merge ``prefix_o.float()*w + current_out*(1-w)``, w = sigmoid(prefix_lse - L) fp32
       (``:326,:302``), cast to fp16 (``:304``).
    Returns [b,M,H,D] fp16."""
    b, M, H, D = q.shape
    N = tree_mask.shape[-1]
    prefix_o, prefix_lse = kvcache_attention(q, k_cache, v_cache, cache_seqlens=cache_lens,
                                             window_size=(window, -1), return_softmax_lse=True)
    ks, vs = [], []
    for i in range(b):
        p = int(cache_lens[i])
        k_cache[i, p + N - M:p + N] = k_new[i]
        v_cache[i, p + N - M:p + N] = v_new[i]
        ks.append(k_cache[i, p:p + N])
        vs.append(v_cache[i, p:p + N])
    k_all = torch.stack(ks, 0)
    v_all = torch.stack(vs, 0)
    cur, L = triton_tree_attention(q.permute(0, 2, 1, 3), k_all.permute(0, 2, 1, 3),
                                   v_all.permute(0, 2, 1, 3), tree_mask)
    weight = torch.sigmoid(prefix_lse - L)                       # b H M fp32   (:326)
    cur = cur.transpose(1, 2)                                    # b M H D fp16
    weight = weight.transpose(1, 2).unsqueeze(-1)                # b M H 1
    out = prefix_o.to(F32) * weight + cur * (1 - weight)         # :302 (fp32)
    return out.to(q.dtype)                                       # :304


# --------------------------------------------------------------------------- #
# a8: RMSNorm   a9: RoPE
# --------------------------------------------------------------------------- #
def rmsnorm(x, weight, eps: float):
    """``LlamaRMSNorm.forward`` (transformers ``modeling_llama``, imported at
    ``longspec/test/llama.py:36``; vendored twin ``longspec/test/qwen2.py:82-87``):
    ``w * dtype( x32 * rsqrt(mean(x32^2) + eps) )``."""
    dt = x.dtype
    x32 = x.to(F32)
    var = x32.pow(2).mean(-1, keepdim=True)
    x32 = x32 * torch.rsqrt(var + eps)
    return weight * x32.to(dt)


def rope_cos_sin(position_ids, inv_freq, attention_scaling: float = 1.0, dtype=F16):
    """``LlamaRotaryEmbedding.forward`` (transformers; vendored twin
    ``longspec/test/qwen2.py:163-178``): freqs = inv_freq (x) pos in fp32,
    emb = cat(freqs, freqs), cos/sin in fp32, * attention_scaling, then ONE cast
    to the activation dtype (G7).  position_ids [b,R] int -> cos, sin [b,R,D]."""
    pos = position_ids.to(F32)
    freqs = pos.unsqueeze(-1) * inv_freq.to(F32).view(1, 1, -1)
    emb = torch.cat((freqs, freqs), dim=-1)
    cos = emb.cos() * attention_scaling
    sin = emb.sin() * attention_scaling
    return cos.to(dtype), sin.to(dtype)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x, cos, sin):
    """``apply_rotary_pos_emb(..., unsqueeze_dim=2)`` (transformers; call sites
    ``longspec/test/llama.py:378``, ``llama_glide.py:294``): x [b,R,heads,D],
    cos/sin [b,R,D]; ``x*cos + rotate_half(x)*sin`` with every product and the
    sum rounded to fp16 (half-split, non-interleaved)."""
    c = cos.unsqueeze(2)
    s = sin.unsqueeze(2)
    return (x * c) + (_rotate_half(x) * s)


# --------------------------------------------------------------------------- #
# a10: accept/reject tree collapse (integer only)
# --------------------------------------------------------------------------- #
def tree_verification(all_spec, all_llm_pred, tree_mask, non_leaf_len: int):
    """``LlamaGlide.tree_verification`` (``longspec/test/llama_glide.py:1128-1175``),
    integer part, restated with explicit loops.

    all_spec / all_llm_pred [b,F] int64, tree_mask [b,F,F] int (row r = ancestors
    of node r incl. itself and the root column 0).  Returns
    (acc_ids [b,acc_max] int64, acc_num [b], double_input [b] int32,
    index_mapping [b,acc_max]).  ``index_mapping`` are the accepted node indices in
    ascending order; the caller moves last-layer KV rows
    ``cache_lens + index_mapping -> cache_lens + [0,acc)`` (``:1159-1173``).
    For batch rows with fewer than acc_max accepted nodes the tail follows the
    reference's ``argsort`` of zeros (``:1153-1154``); only b == 1 is exercised."""
    spec = np.asarray(all_spec)
    pred = np.asarray(all_llm_pred)
    mask = np.asarray(tree_mask)
    b, Fn = spec.shape
    acc_nums, lasts, maps = [], [], []
    for z in range(b):
        father = np.zeros(Fn, dtype=np.int64)
        for r in range(Fn):
            best, bestv = 0, 0
            for c in range(Fn):            # argmax_c((mask - I)[r,c]*c), first max wins (:1136)
                val = (int(mask[z, r, c]) - (1 if r == c else 0)) * c
                if val > bestv:
                    best, bestv = c, val
            father[r] = best
        verify = pred[z, father] == spec[z]                        # :1138
        verify[0] = True                                           # :1139
        final = np.zeros(Fn, dtype=bool)
        for r in range(Fn):                                        # :1140-1141
            final[r] = int((mask[z, r] * verify).sum()) == int(mask[z, r].sum())
        last, lastv = 0, 0
        for r in range(Fn):                                        # :1144
            if int(final[r]) * r > lastv:
                last, lastv = r, int(final[r]) * r
        sel = mask[z, last] != 0                                   # :1147
        acc_nums.append(int(sel.sum()))                            # :1148
        lasts.append(last)
        maps.append(np.nonzero(sel)[0])
    acc_max = max(acc_nums)
    index_mapping = np.zeros((b, acc_max), dtype=np.int64)
    for z in range(b):
        sel_idx = list(maps[z])
        rest = [c for c in range(Fn) if c not in set(sel_idx)]
        full = sel_idx + rest
        index_mapping[z] = np.asarray(full[:acc_max])
    acc_ids = np.take_along_axis(pred, index_mapping, axis=1)      # :1155
    double_input = np.asarray([int(l >= non_leaf_len) for l in lasts], dtype=np.int32)   # :1145
    return (torch.from_numpy(acc_ids), torch.tensor(acc_nums, dtype=torch.int64),
            torch.from_numpy(double_input), torch.from_numpy(index_mapping))


def move_accepted_kv(k_cache, v_cache, cache_lens, index_mapping):
    """KV row move of ``tree_verification`` (``llama_glide.py:1159-1173``): rows
    ``cache_lens + index_mapping[j] -> cache_lens + j`` of the LAST target layer
    only (G6).  Gather-then-scatter (the reference gathers all F rows first), in place."""
    b, n = index_mapping.shape
    for z in range(b):
        L = int(cache_lens[z])
        src = (L + index_mapping[z]).long()
        kk = k_cache[z, src].clone()
        vv = v_cache[z, src].clone()
        k_cache[z, L:L + n] = kk
        v_cache[z, L:L + n] = vv


# --------------------------------------------------------------------------- #
# a11: beam-tree growth (one level)
# --------------------------------------------------------------------------- #
def grow_tree_level(current_logp, history_logp_sum, k: int, base: int):
    """Beam-tree expansion of one level (``llama_glide.py:1046-1067``):
    cumulative log-prob = logp + history; flat top-k over (nodes x vocab);
    ``father = idx // V + base``, ``token = idx % V``.
    current_logp [b,sq,V] fp32, history_logp_sum [b,sq] fp32."""
    b, sq, V = current_logp.shape
    s = current_logp + history_logp_sum[:, :, None]
    top, idx = s.view(b, -1).topk(k, dim=-1)
    return top, idx // V + base, idx % V


# --------------------------------------------------------------------------- #
# e: N-way log-sum-exp merge of partial attention outputs (multi-GPU KV shards)
# --------------------------------------------------------------------------- #
def linear(x, weight, bias=None):
    """``nn.Linear`` as the decode path calls it (q/k/v/o_proj ``longspec/test/llama.py:361-363,390``, the MLP
    and ``lm_head`` ``llama_glide.py:1091``): the exact dot products in fp64, rounded ONCE to the storage
    dtype.  The reference's own GEMM is cuBLAS (third-party, accumulation order unpinned); an fp32-accumulating
    kernel may differ from this by one ulp of the result where the rounding is a near-tie."""
    y = x.double() @ weight.double().t()
    if bias is not None:
        y = y + bias.double()
    return y.to(x.dtype)


def mlp_gate_up(x, gate_weight, up_weight):
    """``act_fn(gate_proj(x)) * up_proj(x)`` (transformers LlamaMLP; vendored ``qwen2.py:229``) with the
    reference's rounding points: each projection, the SiLU and the product are rounded to the storage dtype."""
    g = linear(x, gate_weight).float()
    u = linear(x, up_weight).float()
    s = (g / (1.0 + torch.exp(-g))).to(x.dtype).float()
    return (s * u).to(x.dtype)


def lse_merge(o_parts, lse_parts):
    """N-way generalisation of the reference's 2-way merge
    ``o = o_p*sigmoid(lse_p - lse_t) + o_t*(1 - sigmoid(.))`` (``llama.py:385-387,420``):
    o_parts [W,R,H,D] fp32 (each normalised by its own row sum), lse_parts
    [W,H,R] fp32.  Fixed rank order => deterministic.  Returns (o fp32, lse)."""
    lse = torch.stack(list(lse_parts), 0)                           # W H R
    m = lse.max(dim=0).values
    m_safe = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    w = torch.exp(lse - m_safe)                                     # W H R
    den = w.sum(0)
    o = torch.zeros_like(o_parts[0], dtype=F32)
    for i, op in enumerate(o_parts):
        o = o + op.float() * w[i].transpose(0, 1).unsqueeze(-1)
    o = o / torch.where(den == 0, torch.ones_like(den), den).transpose(0, 1).unsqueeze(-1)
    return o, m_safe + torch.log(den)


# --------------------------------------------------------------------------- #
# temperature > 0: stochastic tree verification
# --------------------------------------------------------------------------- #
class HostDraws:
    """The two random streams ``verify_stochastic`` consumes, in the reference's call order: Python's ``random``
    (``random.choice`` over the remaining children, then ``random.random``) and torch's default generator
    (``torch.multinomial`` of one sample = arg-max of ``p / Exponential(1)`` noise, ATen's fast path)."""

    def choice(self, seq):
        import random
        return random.choice(seq)

    def random(self):
        import random
        return random.random()

    def multinomial(self, p_row):
        return int(torch.multinomial(p_row, num_samples=1).item())


def tree_fathers(tree_mask: torch.Tensor) -> torch.Tensor:
    """Father of every tree node (``llama_glide.py:1188-1191``): the largest ancestor index other than the node itself,
    0 for the root -- ``((mask - I) * arange).argmax(-1)``."""
    b, Fn, _ = tree_mask.shape
    eye = torch.eye(Fn, dtype=tree_mask.dtype)[None]
    return ((tree_mask - eye) * torch.arange(Fn)[None, None, :]).argmax(dim=-1)


def verify_stochastic(input_ids, tree_mask, p_llm, p_ssm, temperature: float, draws=None):
    """``LlamaGlide.verify_stochastic`` (``longspec/test/llama_glide.py:1177-1245``), speculative-sampling walk
    down the draft tree.  input_ids [b,F] int64 (tree node tokens), tree_mask [b,F,F], p_llm [b,F,V] target LOGITS in
    the activation dtype, p_ssm [b,Fs,V] fp32 draft log-probs (Fs >= number of non-leaf nodes), temperature > 0.
    Returns (acc_ids [b, max depth + 2] zero padded, acc_num [b]).

    Reference behaviour kept on purpose (documented):
    * the acceptance ratio of child node ``s`` reads the two distributions at vocabulary index ``s`` -- the child's NODE
      index, not its token id (``:1222``);
    * the target probabilities stay in the activation dtype: ``softmax(logits / T)`` is fp16, ``p + 1e-9`` is a no-op in
      fp16 unless p == 0, every residual update ``max(p - q, 0) / sum`` is rounded to fp16 (``:1231-1235``);
    * ``r <= ratio`` compares in fp32 (Python float against a 0-dim fp32 tensor).
    """
    draws = draws or HostDraws()
    b, Fn, _ = tree_mask.shape
    pl = torch.softmax(p_llm / temperature, dim=-1)                # activation dtype
    ps = torch.softmax(p_ssm / temperature, dim=-1)                # fp32
    fathers = tree_fathers(tree_mask)
    width = int(tree_mask.sum(-1).max()) + 1
    acc_ids = input_ids.new_zeros((b, width))
    acc_num = input_ids.new_zeros(b)
    eps = 1e-9
    for z in range(b):
        path = [int(input_ids[z, 0])]
        cur = 0
        while True:
            kids = [u for u in range(Fn) if u != cur and int(fathers[z, u]) == cur]
            if not kids:
                break
            taken = None
            while kids:
                s = draws.choice(kids)
                r = draws.random()
                ratio = (pl[z, cur, s] + eps) / (ps[z, cur, s] + eps)            # 16-bit / fp32 -> fp32
                if bool(torch.tensor(r, dtype=ratio.dtype) <= ratio):
                    taken = s
                    break
                kids.remove(s)
                row = (pl[z, cur, :] - ps[z, cur, :]).to(pl.dtype)                # residual, rounded to the activation dtype
                row = torch.clamp(row, min=0)
                tot = row.sum()
                if tot > 0:
                    row = row / tot
                pl[z, cur, :] = row
            if taken is None:
                break
            path.append(int(input_ids[z, taken]))
            cur = taken
        path.append(draws.multinomial(pl[z, cur, :]))
        acc_num[z] = len(path)
        acc_ids[z, :len(path)] = torch.tensor(path, dtype=acc_ids.dtype)
    return acc_ids, acc_num


def chain_accept_stochastic(spec_logits, llm_verify_logits, spec_buffer, llm_verify_output, uniform=None, noise=None):
    """The temperature > 0 branch of ``spec_generate`` (``longspec/test/llama_glide.py:715-736``): the greedy chain draft
    ``spec_buffer[:, 1:]`` is accepted position by position with probability ``min(1, p / q)`` at the drafted token, where
    ``q = softmax(spec_logits[:, 1:])`` (fp32) and ``p = softmax(llm_verify_logits[:, :-1])`` in the model dtype (the
    temperature itself is not applied to either -- as in the reference); a rejected position takes one draw from ``p``.
    Returns ``(llm_verify_output, accept_mask)``: ``llm_verify_output[:, :-1]`` rewritten in place as ``:727-731`` does;
    the caller forms ``verification = accept_mask.cumprod(-1)`` (``:732``).
    Random draws, in the reference's order on torch's global generator: ``rand_like(alpha)`` [b, gamma] fp32, then the one
    ``exponential_`` of ``Categorical(p).sample()`` = ``multinomial(p / p.sum, 1)`` = ``argmax(pn / noise)``,
    noise [b * gamma, V] in the model dtype.  ``uniform`` / ``noise`` inject them (tests); default = draw here."""
    b, g1, V = llm_verify_logits.shape
    gamma = g1 - 1
    q_probs = torch.softmax(spec_logits[:, 1:, :], dim=-1)
    p_probs = torch.softmax(llm_verify_logits[:, :-1, :], dim=-1)
    idx = spec_buffer[:, 1:].unsqueeze(-1)
    q_tok = torch.gather(q_probs, -1, idx).squeeze(-1)
    p_tok = torch.gather(p_probs, -1, idx).squeeze(-1)
    eps = 1e-9
    alpha = torch.clip((p_tok + eps) / (q_tok + eps), 0.0, 1.0)
    u = torch.rand_like(alpha) if uniform is None else uniform
    accept = u.lt(alpha)
    p2 = p_probs.reshape(-1, V)
    pn = p2 / p2.sum(-1, keepdim=True)                     # Categorical.__init__ normalises
    nz = torch.empty_like(pn).exponential_(1) if noise is None else noise
    resample = (pn / nz).argmax(dim=-1).reshape(b, gamma)
    llm_verify_output[:, :-1] = torch.where(accept, spec_buffer[:, 1:], resample)
    return llm_verify_output, accept
