"""Target model with exec-type dispatch -- host-side mirror of ``longspec/test/llama.py``.

Same module tree, attribute names (``model.layers[i].self_attn.{K_Cache,V_Cache}``,
``model.embed_tokens``, ``model.rotary_emb``, ``model.norm``, ``lm_head``), call signatures
and KV-cache layout (``[bsz, prompt + max_len, Hkv, D]`` token-major, zero-initialised,
``llama.py:219-222``) as the reference, so ``LlamaGlide`` below it reads like the
reference's.  Every attention / norm / rotary operator goes through ``longspec_amd.ops``
(hand-written HIP kernels behind the C ABI).  Dense projections are ``DecodeLinear``: with at most
80 token rows (every draft / verify / vanilla pass) they stream a pre-packed copy of the weight through
the skinny-GEMM kernel ``ls_linear_fwd``; with more (prefill) they are plain library GEMMs
(``F.linear`` = hipBLASLt through PyTorch-ROCm).

``ops`` is injectable so that the CPU test-suite can drive this host logic with the oracle's
operators; the product default is the HIP operator layer, which raises on CPU tensors.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn


def _default_ops():
    from . import ops
    return ops


# --------------------------------------------------------------------------- #
# rotary embedding
# --------------------------------------------------------------------------- #
def rope_parameters(config):
    """(inv_freq fp32 [D/2], attention_scaling) for the RoPE variants of BASELINE's models:
    'default' (Llama-3-8B-262k theta=283461213, QwQ theta=1e6) and 'linear' (Vicuna-16k x4,
    LongChat x8) -- transformers ``ROPE_INIT_FUNCTIONS`` semantics."""
    dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
    rp = getattr(config, "rope_parameters", None) or {}
    scaling = getattr(config, "rope_scaling", None) or {}
    theta = rp.get("rope_theta", getattr(config, "rope_theta", 10000.0))
    rtype = rp.get("rope_type", scaling.get("rope_type", scaling.get("type", "default")))
    factor = rp.get("factor", scaling.get("factor", 1.0))
    inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    if rtype == "linear":
        inv_freq = inv_freq / factor
    elif rtype != "default":
        raise NotImplementedError(f"rope_type {rtype!r}")
    return inv_freq, 1.0


class LlamaRotaryEmbedding(nn.Module):
    """``LlamaRotaryEmbedding.forward`` (transformers; K9): cos/sin tables in the activation dtype."""

    def __init__(self, config, ops=None):
        super().__init__()
        # plain attribute, not a buffer: model.half() must not round inv_freq to fp16
        self.inv_freq, self.attention_scaling = rope_parameters(config)
        self.ops = ops

    @torch.no_grad()
    def forward(self, x, position_ids):
        if self.inv_freq.device != x.device:
            self.inv_freq = self.inv_freq.to(x.device)
        return self.ops.rope_cos_sin(position_ids, self.inv_freq, self.attention_scaling, x.dtype)


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6, ops=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps
        self.ops = ops

    def forward(self, hidden_states, residual=None):
        """``residual`` given: returns (norm(residual + hidden_states), residual + hidden_states) from one
        kernel -- the `hidden_states = residual + hidden_states` of the decoder layers (``llama.py:492``)
        fused into the norm that follows it (same fp16 rounding of the sum)."""
        if residual is not None:
            return self.ops.rmsnorm(hidden_states, self.weight, self.variance_epsilon, residual=residual)
        return self.ops.rmsnorm(hidden_states, self.weight, self.variance_epsilon)


class DecodeLinear(nn.Linear):
    """``nn.Linear`` (same parameters, same state_dict keys).  A call with <= 80 token rows on the GPU
    streams a packed copy of the weight (MFMA operand order, built once per weight version) through
    ``ops.linear``; anything else is ``F.linear``."""

    def __init__(self, in_features, out_features, bias=True, ops=None):
        super().__init__(in_features, out_features, bias=bias)
        self.ops = ops
        self._packed = {}
        self._packed_key = None

    def packed(self, rope: bool = False):
        """The streamed copy of the weight; ``rope=True``: the q/k layout of ``ops.linear_qkv_rope``."""
        w = self.weight
        key = (w.data_ptr(), w._version, w.dtype, w.device)
        if self._packed_key != key:
            self._packed = {}
            self._packed_key = key
        if rope not in self._packed:
            self._packed[rope] = self.ops.pack_weight(w, rope=True) if rope else self.ops.pack_weight(w)
        return self._packed[rope]

    def streams(self, x) -> bool:
        return self.ops is not None and self.ops.linear_supported(x, self.in_features)

    def forward(self, x, residual=None):
        """``residual``: returns ``residual + linear(x)`` (one launch when the input is decode-shaped)."""
        if self.streams(x):
            return self.ops.linear(x, self.packed(), self.bias, residual=residual)
        y = F.linear(x, self.weight, self.bias)
        return y if residual is None else residual + y


class DecodeEmbedding(nn.Embedding):
    """``nn.Embedding`` (same parameter, same state_dict key).  The token rows of a draft / verify / vanilla
    pass are gathered by ``ops.embed_rows``; prefill-sized inputs use the library kernel."""

    def __init__(self, num_embeddings, embedding_dim, ops=None):
        super().__init__(num_embeddings, embedding_dim)
        self.ops = ops

    def forward(self, input_ids):
        if self.ops is not None and self.ops.embed_supported(input_ids, self.weight):
            return self.ops.embed_rows(self.weight, input_ids)
        return F.embedding(input_ids, self.weight)


class LlamaMLP(nn.Module):
    """``down_proj(act_fn(gate_proj(x)) * up_proj(x))`` (transformers LlamaMLP; vendored qwen2.py:218-230).
    Decode-shaped calls run gate|up + SiLU + product as ONE weight-streaming launch."""

    def __init__(self, config, ops=None):
        super().__init__()
        bias = getattr(config, "mlp_bias", False)
        self.ops = ops
        self.gate_proj = DecodeLinear(config.hidden_size, config.intermediate_size, bias=bias, ops=ops)
        self.up_proj = DecodeLinear(config.hidden_size, config.intermediate_size, bias=bias, ops=ops)
        self.down_proj = DecodeLinear(config.intermediate_size, config.hidden_size, bias=bias, ops=ops)
        self._gate_up = None
        self._gate_up_key = None

    def _packed_gate_up(self):
        g, u = self.gate_proj.weight, self.up_proj.weight
        key = (g.data_ptr(), g._version, u.data_ptr(), u._version, g.dtype, g.device)
        if self._gate_up_key != key:
            self._gate_up = self.ops.pack_gate_up(g, u)
            self._gate_up_key = key
        return self._gate_up

    def forward(self, x, residual=None):
        """``residual``: returns ``residual + mlp(x)``, the add riding in down_proj's epilogue for decode-shaped inputs."""
        if self.gate_proj.bias is None and self.gate_proj.streams(x) and self.gate_proj.out_features % 16 == 0:
            return self.down_proj(self.ops.mlp_gate_up(x, self._packed_gate_up()), residual=residual)
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x), residual=residual)


FUSE_QKV_ROPE = True      # decode-shaped q|k|v projections rotate q and k in the GEMM epilogue (tools/ab_round.py flips it)


def project_qkv(ops, x, q_proj, k_proj, v_proj, position_embeddings, num_heads, num_kv_heads, head_dim=128):
    """q/k/v projections + rotary embedding of q and k (``llama.py:371-378``): [bsz, q_len, heads, head_dim] each.
    Decode-shaped input: ONE launch over the three packed weights with the rotation in its epilogue (the outputs are
    column slices of one [rows, Nq+Nk+Nv] buffer); otherwise three plain linears and ``rope_apply_``.
    ``k_proj`` None: q only (the draft's cross-attention)."""
    bsz, q_len, _ = x.shape
    cos, sin = position_embeddings
    projs = [p for p in (q_proj, k_proj, v_proj) if p is not None]
    streams = q_proj.streams(x) and all(m.out_features % 128 == 0 for m in projs)
    fused = streams and head_dim == 128 and FUSE_QKV_ROPE
    if fused:
        packed = [m.packed(rope=i < 2) for i, m in enumerate(projs)]
        outs = ops.linear_qkv_rope(x, packed, [m.bias for m in projs], cos, sin)
    elif streams:
        outs = ops.linear_multi(x, [m.packed() for m in projs], [m.bias for m in projs])
    else:
        outs = [m(x) for m in projs]
    q = outs[0].view(bsz, q_len, num_heads, head_dim)
    k = outs[1].view(bsz, q_len, num_kv_heads, head_dim) if k_proj is not None else q[:, :, :0]
    v = outs[2].view(bsz, q_len, num_kv_heads, head_dim) if v_proj is not None else None
    if not fused:
        ops.rope_apply_(q, k, cos, sin)
    return q, k, v


def chunked_causal_prefill(ops, q, k, v, k_cache, v_cache, window_left=-1):
    """Prompt attention (K13, ``flash_attn_func(causal=True[, window_size=(512,-1)])``,
    ``llama.py:218`` / ``llama_glide.py:227``) + cache fill, delegated to the operator layer."""
    return ops.prefill_attention(q, k, v, k_cache, v_cache, window_left=window_left)


class LlamaAttention(nn.Module):
    """``LlamaAttention`` (``longspec/test/llama.py:55-421``): prefill / decoding / tree_decoding."""

    QKV_BIAS = None      # None: config.attention_bias (Llama); True in the Qwen2 twin (qwen2.py:214-216)

    def __init__(self, config, layer_idx: int, ops=None):
        super().__init__()
        self.config = config
        self.layer_idx = layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = getattr(config, "head_dim", None) or self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        bias = getattr(config, "attention_bias", False) if self.QKV_BIAS is None else self.QKV_BIAS
        self.q_proj = DecodeLinear(self.hidden_size, self.num_heads * self.head_dim, bias=bias, ops=ops)
        self.k_proj = DecodeLinear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias, ops=ops)
        self.v_proj = DecodeLinear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias, ops=ops)
        self.o_proj = DecodeLinear(self.num_heads * self.head_dim, self.hidden_size, bias=False, ops=ops)
        self.K_Cache = None
        self.V_Cache = None
        self.max_len = 512
        self.last_layer = (config.num_hidden_layers == layer_idx + 1)
        self.softmax_scale = 1 / (128 ** 0.5)          # hard-coded in the reference (llama.py:95, G1)
        self.ops = ops
        self.kv_len_hint = None                          # host-side upper bound of cache_lens (grid sizing)
        self.timing = None                               # optional callable -> (start, stop) events (bench.py)
        self.xchg_timing = None                          # the same around the exchange + merge of a sharded call
        self.shard = None                                # dist.KVShard when the prefix KV is sequence-sharded

    def _qkv(self, hidden_states, position_embeddings):
        return project_qkv(self.ops, hidden_states, self.q_proj, self.k_proj, self.v_proj, position_embeddings,
                           self.num_heads, self.num_key_value_heads, self.head_dim)

    def forward(self, hidden_states, position_embeddings, cache_lens=None, flex_attn=None, tree_mask=None,
                exec_type="training", induction_head=False, tree_mask_bits=None):
        if exec_type in ("prefill", "prefill_torch"):            # the *_torch twins of the reference (llama.py:132-197)
            y = self.prefill(hidden_states, position_embeddings)   # compute the same function with eager attention
        elif exec_type in ("decoding", "decoding_torch"):
            y = self.decoding(hidden_states, position_embeddings, cache_lens)
        elif exec_type == "magicdec_prefill":
            y = self.magicdec_prefill(hidden_states, position_embeddings)
        elif exec_type == "magicdec_decoding":
            y = self.fix_stream_spec(hidden_states, position_embeddings, cache_lens)
        elif exec_type == "tree_decoding":
            y = self.tree_decoding(hidden_states, position_embeddings, cache_lens, tree_mask, tree_mask_bits)
        else:
            raise ValueError(f"Unknown inference_type: {exec_type}")
        return y, None

    def prefill(self, hidden_states, position_embeddings):          # llama.py:199-226
        if self.shard is not None and self.shard.prefill_ctx is not None:
            return self.sharded_prefill(hidden_states, position_embeddings)
        bsz, q_len, _ = hidden_states.size()
        q, k, v = self._qkv(hidden_states, position_embeddings)
        self.K_Cache = q.new_zeros((bsz, q_len + self.max_len, self.num_key_value_heads, self.head_dim))
        self.V_Cache = q.new_zeros((bsz, q_len + self.max_len, self.num_key_value_heads, self.head_dim))
        attn = chunked_causal_prefill(self.ops, q, k, v, self.K_Cache, self.V_Cache)
        return self.o_proj(attn.reshape(bsz, q_len, -1))

    def sharded_prefill(self, hidden_states, position_embeddings):
        """Prefill of this rank's slice of the prompt: ``hidden_states`` are the local rows
        [lo, lo + n).  Every rank's K/V rows of the layer are all-gathered into a scratch cache laid out by global row, and
        the local rows are appended onto rows [0, lo) of it with the SAME causal block-wise calls the single-GPU prefill
        makes for those rows (chunk size from the whole prompt, boundaries at global multiples of it: the same numbers for
        every chunk a shard boundary does not cut).  What stays is the local shard:
        K_Cache/V_Cache = [local rows | room for the tail], exactly what ``dist.shard_model_kv`` leaves behind."""
        sh = self.shard
        lo, n, P = sh.prefill_ctx
        bsz, q_len, _ = hidden_states.size()
        assert q_len == n and 0 < n <= sh.Ls, "sharded prefill: every rank holds between 1 and shard_rows prompt rows"
        q, k, v = self._qkv(hidden_states, position_embeddings)
        Hkv, D = self.num_key_value_heads, self.head_dim
        pad_k, pad_v = k.new_zeros((bsz, sh.Ls, Hkv, D)), v.new_zeros((bsz, sh.Ls, Hkv, D))
        pad_k[:, :n], pad_v[:, :n] = k, v
        # [W, bsz, Ls, Hkv, D] -> [bsz, W*Ls, Hkv, D]: global row = rank * Ls + i (every rank before the tail is full)
        all_k = sh.gather_rows(pad_k).permute(1, 0, 2, 3, 4).reshape(bsz, sh.world * sh.Ls, Hkv, D)
        all_v = sh.gather_rows(pad_v).permute(1, 0, 2, 3, 4).reshape(bsz, sh.world * sh.Ls, Hkv, D)
        attn = self.ops.prefill_attention(q, k, v, all_k, all_v, start=lo, total=P)
        rows = (n if sh.is_tail else sh.Ls) + self.max_len
        self.K_Cache = q.new_zeros((bsz, rows, Hkv, D))
        self.V_Cache = q.new_zeros((bsz, rows, Hkv, D))
        self.K_Cache[:, :n], self.V_Cache[:, :n] = k, v
        return self.o_proj(attn.reshape(bsz, q_len, -1))

    STREAM_SINK, STREAM_WINDOW = 32, 1024      # StreamingLLM cache of the MagicDec baseline (llama.py:255-262)

    def magicdec_prefill(self, hidden_states, position_embeddings):       # llama.py:228-264
        """Normal prefill, then the drafter's streaming cache: rows [0,32) = the first 32 prompt rows (attention
        sinks), rows [32,1056) = the last 1024 prompt rows; generated rows are appended behind them."""
        y = self.prefill(hidden_states, position_embeddings)
        bsz, q_len, _ = hidden_states.size()
        sink, win = self.STREAM_SINK, self.STREAM_WINDOW
        if q_len < win:
            raise ValueError(f"magicdec needs a prompt of at least {win} tokens, got {q_len}")    # the reference's slice assignment fails there too
        shape = (bsz, win + sink + self.max_len, self.num_key_value_heads, self.head_dim)
        self.stream_k_cache = self.K_Cache.new_zeros(shape)
        self.stream_v_cache = self.K_Cache.new_zeros(shape)
        for dst, src in ((self.stream_k_cache, self.K_Cache), (self.stream_v_cache, self.V_Cache)):
            dst[:, :sink] = src[:, :sink]
            dst[:, sink:sink + win] = src[:, q_len - win:q_len]
        return y

    def fix_stream_spec(self, hidden_states, position_embeddings, cache_lens):   # llama.py:331-355
        bsz, q_len, _ = hidden_states.size()
        q, k, v = self._qkv(hidden_states, position_embeddings)
        attn = self.ops.kvcache_attention(q, self.stream_k_cache, self.stream_v_cache, k, v, causal=True,
                                          cache_seqlens=cache_lens, kv_len_hint=self.kv_len_hint)
        return self.o_proj(attn.view(bsz, q_len, self.hidden_size))

    def decoding(self, hidden_states, position_embeddings, cache_lens):   # llama.py:304-329
        bsz, q_len, _ = hidden_states.size()
        q, k, v = self._qkv(hidden_states, position_embeddings)
        attn = self.ops.kvcache_attention(q, self.K_Cache, self.V_Cache, k, v, causal=True, cache_seqlens=cache_lens,
                                          kv_len_hint=self.kv_len_hint)
        return self.o_proj(attn.view(bsz, q_len, self.hidden_size))

    def tree_decoding(self, hidden_states, position_embeddings, cache_lens, tree_mask=None, tree_mask_bits=None):
        """llama.py:357-392: prefix flash-decoding + tree part + fp16 merge, one fused op."""
        q, k, v = self._qkv(hidden_states, position_embeddings)
        return self.tree_attend(q, k, v, cache_lens, tree_mask, tree_mask_bits, hidden_states.dtype)

    def tree_attend(self, q, k, v, cache_lens, tree_mask=None, tree_mask_bits=None, dtype=None):
        """The attention of ``tree_decoding`` on projected, rotated q/k/v [bsz, q_len, heads, 128], then o_proj."""
        bsz, q_len = q.shape[0], q.shape[1]
        hidden_states = SimpleNamespace(dtype=dtype if dtype is not None else q.dtype)
        o_proj = self.o_proj
        if tree_mask is None and tree_mask_bits is None:
            assert q_len == 1, "You are in the first step of tree decoding, thus you should not input qlen > 2 without tree mask"
            attn = self.ops.kvcache_attention(q, self.K_Cache, self.V_Cache, k, v, cache_seqlens=cache_lens, causal=True,
                                              kv_len_hint=self.kv_len_hint)
        else:
            if tree_mask_bits is None:
                tree_mask_bits = self.ops.pack_tree_mask(tree_mask)
            if self.shard is not None:
                sh = self.shard
                extra = {"timing": self.timing()} if self.timing is not None else {}
                call = self.ops.sharded_verify_attention(q, k, v, self.K_Cache, self.V_Cache, sh.pass_len(cache_lens),
                                                         tree_mask_bits, self.last_layer, softmax_scale=self.softmax_scale,
                                                         kv_len_hint=sh.local_hint(self.kv_len_hint), **extra)
                if self.xchg_timing is not None:
                    call.xchg_timing = self.xchg_timing()
                attn = sh.attend(call)
                return o_proj(attn.view(bsz, q_len, self.hidden_size).to(hidden_states.dtype))
            extra = {"timing": self.timing()} if self.timing is not None else {}
            attn = self.ops.verify_attention(q, k, v, self.K_Cache, self.V_Cache, cache_lens, tree_mask_bits,
                                             self.last_layer, softmax_scale=self.softmax_scale,
                                             kv_len_hint=self.kv_len_hint, **extra)
        return o_proj(attn.view(bsz, q_len, self.hidden_size).to(hidden_states.dtype))


class LlamaDecoderLayer(nn.Module):
    ATTENTION_CLS = LlamaAttention

    def __init__(self, config, layer_idx: int, ops=None):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = self.ATTENTION_CLS(config, layer_idx, ops=ops)
        self.mlp = LlamaMLP(config, ops=ops)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps, ops=ops)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps, ops=ops)

    def forward(self, hidden_states, position_embeddings, cache_lens=None, flex_attn=None, exec_type=None, tree_mask=None,
                induction_head=False, tree_mask_bits=None, pending_residual=None, defer_residual=False, prenormed=None):
        """``pending_residual``: the previous layer's MLP output has not been added to its residual stream yet;
        the sum (rounded to the storage dtype exactly like ``residual + hidden_states``, llama.py:492) is formed
        inside this layer's first norm kernel.  ``defer_residual``: return (mlp_out, residual) un-added for the
        next norm to fuse.  ``prenormed``: ``input_layernorm(hidden_states)`` already computed by the pass's head launch
        (``ops.pass_head``; first layer only)."""
        if prenormed is not None:
            residual = hidden_states
            hidden_states = prenormed
        elif pending_residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual=pending_residual)
        hidden_states, kv_cache = self.self_attn(hidden_states=hidden_states, position_embeddings=position_embeddings,
                                                 cache_lens=cache_lens, exec_type=exec_type, tree_mask=tree_mask,
                                                 tree_mask_bits=tree_mask_bits)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual=residual)   # residual + attn, then norm
        hidden_states = self.mlp(hidden_states)
        if defer_residual:
            return hidden_states, residual
        hidden_states = residual + hidden_states
        return hidden_states, kv_cache


class LlamaModel(nn.Module):
    LAYER_CLS = LlamaDecoderLayer

    def __init__(self, config, ops=None):
        super().__init__()
        self.config = config
        self.ops = ops
        self.padding_idx = getattr(config, "pad_token_id", None)
        self.vocab_size = config.vocab_size
        self.embed_tokens = DecodeEmbedding(config.vocab_size, config.hidden_size, ops=ops)
        self.layers = nn.ModuleList([self.LAYER_CLS(config, i, ops=ops) for i in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps, ops=ops)
        self.rotary_emb = LlamaRotaryEmbedding(config, ops=ops)

    def forward(self, input_ids, position_ids=None, position_embeddings=None, inputs_embeds=None, cache_lens=None,
                flex_attn=None, exec_type=None, tree_mask=None, induction_head=False, tree_mask_bits=None):
        """``tree_mask_bits`` (with ``position_ids``) is the packed form of ``tree_mask`` from ``ops.tree_verify_inputs``;
        given both, the dense mask is not needed."""
        shard = getattr(self.layers[0].self_attn, "shard", None)
        if shard is not None:
            shard.begin_pass()
        # decode-shaped passes (<= 128 token rows): embedding gather + RoPE table + the first layer's input norm are ONE launch
        # (ops.pass_head, bit-identical to the three operators; the positions `arange + cache_lens` are formed inside it)
        head = (inputs_embeds is None and position_embeddings is None and input_ids is not None and input_ids.dim() == 2
                and cache_lens is not None and (position_ids is not None or tree_mask is None)
                and getattr(self.ops, "pass_head", None) is not None and getattr(self.ops, "PASS_HEAD", True)
                and self.ops.pass_head_supported(input_ids, self.embed_tokens.weight, self.layers[0].input_layernorm.weight))
        if position_ids is None and not head:                       # llama.py:571-577
            if tree_mask is None:
                position_ids = torch.arange(0, input_ids.size(1), device=input_ids.device)[None, :]
                if cache_lens is not None:
                    position_ids = position_ids + cache_lens[:, None]
            else:
                position_ids = self.ops.tree_positions(tree_mask, cache_lens)
        if tree_mask is not None and tree_mask_bits is None:
            tree_mask_bits = self.ops.pack_tree_mask(tree_mask)     # once per pass, shared by all layers
        prenormed = None
        if head:
            rot, ln = self.rotary_emb, self.layers[0].input_layernorm
            if rot.inv_freq.device != input_ids.device:
                rot.inv_freq = rot.inv_freq.to(input_ids.device)
            inputs_embeds, prenormed, position_embeddings = self.ops.pass_head(
                self.embed_tokens.weight, input_ids, rot.inv_freq, rot.attention_scaling, ln.weight, ln.variance_epsilon,
                position_ids=position_ids, pos_base=cache_lens if position_ids is None else None)
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        hidden_states = inputs_embeds
        if position_embeddings is None:
            position_embeddings = self.rotary_emb(hidden_states, position_ids)
        residual = None                                            # every `residual + mlp(x)` rides in the next norm kernel
        for decoder_layer in self.layers:
            hidden_states, residual = decoder_layer(hidden_states, position_embeddings, cache_lens, flex_attn, exec_type,
                                                    tree_mask, induction_head, tree_mask_bits=tree_mask_bits,
                                                    pending_residual=residual, defer_residual=True, prenormed=prenormed)
            prenormed = None
        hidden_states, _ = self.norm(hidden_states, residual=residual)
        return SimpleNamespace(last_hidden_state=hidden_states, past_key_values=None)

    def set_kv_len_hint(self, hint: Optional[int]):
        for layer in self.layers:
            layer.self_attn.kv_len_hint = hint


class LlamaForCausalLM(nn.Module):
    MODEL_CLS = LlamaModel

    def __init__(self, config, ops=None):
        super().__init__()
        self.config = config
        self.ops = ops if ops is not None else _default_ops()
        self.model = self.MODEL_CLS(config, ops=self.ops)
        self.vocab_size = config.vocab_size
        self.lm_head = DecodeLinear(config.hidden_size, config.vocab_size, bias=False, ops=self.ops)

    def set_max_gen_len(self, max_gen_len):                          # llama.py:646-648
        for layer in self.model.layers:
            layer.self_attn.max_len = max_gen_len
