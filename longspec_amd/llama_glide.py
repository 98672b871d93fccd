"""Draft ("glide") layer and the speculative-decoding loops -- host-side mirror of
``longspec/test/llama_glide.py``: same classes, method names, argument meaning and
return tuples, same draft-checkpoint tensor names (20 tensors, q/k/v with
bias), driven by the HIP operator layer.

Differences from the reference that are deliberate and invisible to callers:
* every attention / norm / rotary / tree-collapse operator is a hand-written HIP kernel
  (``longspec_amd.ops``); there is no flash-attn, Triton or torch.compile dependency;
* the round loop reads ONE scalar per round from the device (``acc_num``) instead of the
  reference's several implicit synchronisations (``llama_glide.py:1069-1089,1118-1121,1149``);
* the host keeps an upper bound of every cache length so kernels can size their grids
  without reading device memory.
"""
from __future__ import annotations

import time
from typing import List, Optional

import torch
from torch import nn

from .llama import (DecodeLinear, LlamaForCausalLM, LlamaMLP, LlamaRMSNorm, chunked_causal_prefill,
                    project_qkv, _default_ops)


class GlideAttention(nn.Module):
    """``GlideAttention`` (``llama_glide.py:23-385``): draft self-attention with a sliding
    window of 512 over its own KV cache, and cross-attention over the target's last-layer KV."""

    WINDOW = 512
    CACHE_PAD = 128      # extra draft-cache rows at prefill (llama_glide.py:218-219); 0 in the Qwen2 twin (qwen2_glide.py:217-218)

    def __init__(self, config, layer_idx: Optional[int] = None, ops=None):
        super().__init__()
        self.config = config
        self.layer_idx = layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.q_proj = DecodeLinear(self.hidden_size, self.num_heads * self.head_dim, bias=True, ops=ops)     # bias=True even for Llama (:49-52)
        self.k_proj = DecodeLinear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=True, ops=ops)
        self.v_proj = DecodeLinear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=True, ops=ops)
        self.o_proj = DecodeLinear(self.num_heads * self.head_dim, self.hidden_size, bias=False, ops=ops)
        self.K_Cache = None
        self.V_Cache = None
        self.max_len = 512
        self.softmax_scale = 1 / (self.head_dim ** 0.5)
        self.ops = ops
        self.kv_len_hint = None          # bound of the draft cache length
        self.llm_kv_len_hint = None      # bound of the target last-layer KV length
        self.shard = None                # dist.KVShard: the target KV this layer cross-attends is sequence-sharded

    def forward(self, hidden_states, position_embeddings, cache_lens=None, flex_attn=None, exec_type="training",
                k_cache=None, v_cache=None, llm_kv_len=None, tree_mask=None, tree_mask_bits=None):
        if exec_type in ["prefill", "sa_prefill"]:
            y = self.prefill(hidden_states, position_embeddings)
        elif exec_type == "sa_decoding":
            y = self.decoding(hidden_states, position_embeddings, cache_lens, K_Cache=None, V_Cache=None)
        elif exec_type in ["decoding", "ca_decoding", "ca_prefill"]:
            y = self.decoding(hidden_states, position_embeddings, cache_lens, k_cache, v_cache, llm_kv_len)
        elif exec_type in ["sa_tree_decoding"]:
            y = self.tree_decoding(hidden_states, position_embeddings, cache_lens, None, None, llm_kv_len, tree_mask, tree_mask_bits)
        elif exec_type in ["ca_tree_decoding"]:
            y = self.tree_decoding(hidden_states, position_embeddings, cache_lens, k_cache, v_cache, llm_kv_len, tree_mask, tree_mask_bits)
        else:
            raise ValueError(f"Unknown inference_type: {exec_type}")
        return y

    def _qkv(self, hidden_states, position_embeddings, need_kv=True):
        if need_kv:
            return project_qkv(self.ops, hidden_states, self.q_proj, self.k_proj, self.v_proj, position_embeddings,
                               self.num_heads, self.num_key_value_heads, self.head_dim)
        # cross-attention: the reference projects k/v it never uses (:248-249 vs :265); skip them
        q, _, _ = project_qkv(self.ops, hidden_states, self.q_proj, None, None, position_embeddings,
                              self.num_heads, self.num_key_value_heads, self.head_dim)
        return q, None, None

    def _cross(self, q, K_Cache, V_Cache, llm_kv_len, causal: bool):
        """Cross-attention over the target's last-layer KV (K7), whole or sequence-sharded."""
        if self.shard is None:
            return self.ops.kvcache_attention(q, K_Cache, V_Cache, causal=causal, cache_seqlens=llm_kv_len.int(),
                                              kv_len_hint=self.llm_kv_len_hint)
        sh = self.shard
        # bottom-right causal alignment only matters on the rank that owns the tail of the sequence
        call = self.ops.sharded_prefix_attention(q, K_Cache, V_Cache, sh.pass_len(llm_kv_len),
                                                 causal=causal and sh.is_tail, kv_len_hint=sh.local_hint(self.llm_kv_len_hint))
        return sh.attend(call)

    def prefill(self, hidden_states, position_embeddings):                      # :206-233
        bsz, q_len, _ = hidden_states.size()
        q, k, v = self._qkv(hidden_states, position_embeddings)
        self.K_Cache = q.new_zeros((bsz, q_len + self.max_len + self.CACHE_PAD, self.num_key_value_heads, self.head_dim))
        self.V_Cache = q.new_zeros((bsz, q_len + self.max_len + self.CACHE_PAD, self.num_key_value_heads, self.head_dim))
        attn = chunked_causal_prefill(self.ops, q, k, v, self.K_Cache, self.V_Cache, window_left=self.WINDOW)
        return self.o_proj(attn.reshape(bsz, q_len, self.hidden_size))

    def prefill_cache_only(self, hidden_states, position_embeddings):
        """What the draft's prefill leaves behind: the K/V rows of the prompt.  The layer's outputs are discarded by
        every caller (:968-975), and these rows are a row-wise function of the token ids -- so the sequence-sharded path,
        where no rank holds the whole target KV the cross-attention would read, fills the cache with just this."""
        bsz, q_len, _ = hidden_states.size()
        _, k, v = self._qkv(hidden_states, position_embeddings)
        shape = (bsz, q_len + self.max_len + self.CACHE_PAD, self.num_key_value_heads, self.head_dim)
        self.K_Cache, self.V_Cache = k.new_zeros(shape), v.new_zeros(shape)
        self.K_Cache[:, :q_len], self.V_Cache[:, :q_len] = k, v

    def decoding(self, hidden_states, position_embeddings, cache_lens, K_Cache, V_Cache, llm_kv_len=None):   # :235-270
        bsz, q_len, _ = hidden_states.size()
        if K_Cache is None:
            q, k, v = self._qkv(hidden_states, position_embeddings)
            attn = self.ops.kvcache_attention(q, self.K_Cache, self.V_Cache, k, v, window_size=(self.WINDOW, -1), causal=True,
                                              cache_seqlens=cache_lens.int(), kv_len_hint=self.kv_len_hint)
        else:
            q, _, _ = self._qkv(hidden_states, position_embeddings, need_kv=False)
            attn = self._cross(q, K_Cache, V_Cache, llm_kv_len, causal=True)
        return self.o_proj(attn.view(bsz, q_len, self.hidden_size))

    def tree_decoding(self, hidden_states, position_embeddings, cache_lens, K_Cache, V_Cache, llm_kv_len=None,
                      tree_mask=None, tree_mask_bits=None):                     # :272-329
        bsz, q_len, _ = hidden_states.size()
        if K_Cache is not None:
            q, _, _ = self._qkv(hidden_states, position_embeddings, need_kv=False)
            attn = self._cross(q, K_Cache, V_Cache, llm_kv_len, causal=False)
        else:
            q, k, v = self._qkv(hidden_states, position_embeddings)
            if tree_mask_bits is None:
                tree_mask_bits = self.ops.pack_tree_mask(tree_mask)
            attn = self.ops.draft_tree_attention(q, k, v, self.K_Cache, self.V_Cache, cache_lens, tree_mask_bits,
                                                 tree_mask.size(-1), window=self.WINDOW, kv_len_hint=self.kv_len_hint)
        return self.o_proj(attn.view(bsz, q_len, self.hidden_size).to(hidden_states.dtype))


class LlamaGlideDecoderLayer(nn.Module):
    """``LlamaGlideDecoderLayer`` (``llama_glide.py:388-468``): norm -> self-attn -> +res ->
    norm -> cross-attn -> +res -> norm -> MLP -> +res; no final norm."""

    ATTENTION_CLS = GlideAttention

    def __init__(self, config, ops=None):
        super().__init__()
        self.config = config
        self.ops = ops if ops is not None else _default_ops()
        self.hidden_size = config.hidden_size
        self.layer_idx = 0
        self.self_attn = self.ATTENTION_CLS(config, self.layer_idx, ops=self.ops)
        self.cross_attn = self.ATTENTION_CLS(config, self.layer_idx, ops=self.ops)
        self.mlp = LlamaMLP(config, ops=self.ops)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps, ops=self.ops)
        self.post_self_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps, ops=self.ops)
        self.post_cross_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps, ops=self.ops)

    def set_max_gen_len(self, max_gen_len):
        self.self_attn.max_len = max_gen_len

    def prefill_cache_only(self, hidden_states, position_embeddings):
        self.self_attn.prefill_cache_only(self.input_layernorm(hidden_states), position_embeddings)

    def forward(self, hidden_states, position_embeddings, llm_kv, cache_lens=None, exec_type=None, llm_kv_len=None,
                tree_mask=None, tree_mask_bits=None, prenormed=None):
        """``tree_mask_bits``: the packed ``tree_mask`` when the caller already has it (``ops.tree_grow``);
        ``prenormed``: ``input_layernorm(hidden_states)`` when the pass's head launch (``ops.pass_head``) already formed it."""
        bits = tree_mask_bits
        if bits is None and tree_mask is not None:
            bits = self.ops.pack_tree_mask(tree_mask)
        if self.cross_attn.shard is not None:
            self.cross_attn.shard.begin_pass()
        residual = hidden_states
        hidden_states = self.input_layernorm(hidden_states) if prenormed is None else prenormed
        hidden_states = self.self_attn(hidden_states=hidden_states, position_embeddings=position_embeddings,
                                       cache_lens=cache_lens, exec_type="sa_" + exec_type, tree_mask=tree_mask,
                                       tree_mask_bits=bits)
        hidden_states, residual = self.post_self_attention_layernorm(hidden_states, residual=residual)   # + residual, norm
        hidden_states = self.cross_attn(hidden_states=hidden_states, position_embeddings=position_embeddings,
                                        cache_lens=cache_lens, exec_type="ca_" + exec_type, k_cache=llm_kv[0],
                                        v_cache=llm_kv[1], llm_kv_len=llm_kv_len, tree_mask=tree_mask, tree_mask_bits=bits)
        hidden_states, residual = self.post_cross_attention_layernorm(hidden_states, residual=residual)  # + residual, norm
        return self.mlp(hidden_states, residual=residual)           # `hidden_states = residual + mlp(...)` (:466)


def _sync(t: torch.Tensor):
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


class LlamaGlide(LlamaForCausalLM):
    """``LlamaGlide`` (``llama_glide.py:471-1245``).  ``LlamaGlide(config, target_model_path,
    glide_path=None)`` -- the reference's signature -- loads the target and the draft checkpoint (local
    directories or hub ids, safetensors / .bin, sharded or not) in fp16 onto the current GPU, as the
    reference's ``from_pretrained(..., torch_dtype=torch.float16, device_map="auto")`` does (``:474,480``);
    ``config`` may be a ``transformers`` config object or any attribute namespace.
    ``target_model_path=None`` builds random-init modules (tests, synthetic benchmarks).  ``device`` /
    ``dtype`` / ``ops`` are extensions: with the default HIP operator layer the model lives on
    ``torch.cuda.current_device()``; an injected ``ops`` (the CPU oracle of the host-logic tests) keeps
    it where it is."""

    GLIDE_LAYER_CLS = LlamaGlideDecoderLayer

    def __init__(self, config, target_model_path=None, glide_path=None, ops=None, dtype=torch.float16, device=None):
        super().__init__(config, ops=ops)
        self.glide = self.GLIDE_LAYER_CLS(config, ops=self.ops)
        if target_model_path is not None or glide_path is not None:
            from .checkpoint import load_draft_checkpoint, load_target_checkpoint
            if target_model_path is not None:
                load_target_checkpoint(self, target_model_path)
            if glide_path is not None:
                load_draft_checkpoint(self.glide, glide_path)
        self.to(dtype)
        if device is None and ops is None and torch.cuda.is_available():
            device = torch.device("cuda", torch.cuda.current_device())
        if device is not None:
            self.to(device)
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    GRAPH_ROUNDS = True      # capture tree rounds into HIP graphs (one per accepted-token count), see tree_round
    GRAPH_AFTER = 256        # ... once a generation has run this many rounds: a capture costs ~10 ms and a replay saves
                             # <= 0.1 ms on a fast host (more on a slow or busy one), so only long generations
                             # (LongSpec's long-CoT case) pay it back; benchmarks capture up front (prepare_tree_graphs)
    GRAPH_TIER = 4096        # a captured round / step is sized (host-side bounds of the KV lengths -> launch grids, split
                             # counts, kernel choice) for the tokens emitted so far rounded up to the next multiple of this;
                             # a generation that outgrows its tier drops its graphs and captures the next tier's (a 20 000-token
                             # long-CoT generation: five tiers, ~12 captures of ~10 ms).  Sizing every graph for the whole
                             # budget instead made a 1k-token prompt with a 20k budget run 21k-row launch shapes from round 1.

    # ------------------------------------------------------------------------------------------
    def set_max_gen_len(self, max_gen_len):
        super().set_max_gen_len(max_gen_len)

    def _set_hints(self, target_bound: Optional[int], draft_bound: Optional[int]):
        """Host-side upper bounds of the KV lengths the next kernels will see."""
        self.model.set_kv_len_hint(target_bound)
        self.glide.self_attn.kv_len_hint = draft_bound
        self.glide.cross_attn.llm_kv_len_hint = target_bound

    def _draft_head(self, ids, position_ids=None, pos_base=None, pos_add: int = 0):
        """Head of a draft pass: ``embed_tokens(ids)``, the RoPE table of its positions (``position_ids`` [b,q] int64, or
        ``arange(q) + pos_base[:, None] + pos_add``) and the draft layer's ``input_layernorm`` (``llama_glide.py:1003-1006,
        1030-1033,437``).  One launch on the HIP operator layer (``ops.pass_head``), three operators otherwise (the CPU oracle
        of the host-logic tests); returns (embeds, normed or None, position_embeddings)."""
        ops, ln, rot = self.ops, self.glide.input_layernorm, self.model.rotary_emb
        if (getattr(ops, "pass_head", None) is not None and getattr(ops, "PASS_HEAD", True)
                and ops.pass_head_supported(ids, self.model.embed_tokens.weight, ln.weight)):
            if rot.inv_freq.device != ids.device:
                rot.inv_freq = rot.inv_freq.to(ids.device)
            return ops.pass_head(self.model.embed_tokens.weight, ids, rot.inv_freq, rot.attention_scaling, ln.weight,
                                 ln.variance_epsilon, position_ids=position_ids, pos_base=pos_base, pos_add=pos_add)
        hidden_states = self.model.embed_tokens(ids)
        if position_ids is None:
            position_ids = torch.arange(ids.size(1), device=ids.device)[None, :] + pos_base[:, None] + pos_add
        return hidden_states, None, self.model.rotary_emb(hidden_states, position_ids)

    def _stop_id(self, eos_id, loop: str):
        """Token whose appearance ends a loop.  The Llama twin tests ``self.config.eos_token_id`` in all
        three loops (llama_glide.py:578,767,1120) and ignores the ``eos_id`` argument for that."""
        return getattr(self.config, "eos_token_id", None)

    def _tree_output_fill(self, eos_id):
        """Initial content of ``output_ids`` in tree_spec_generate (llama_glide.py:937: eos_id, G8)."""
        return eos_id

    def _last_kv(self):
        attn = self.model.layers[-1].self_attn
        return attn.K_Cache, attn.V_Cache

    # ------------------------------------------------------------------------------------------
    def _clear_shard(self):
        """A shard set by an earlier ``tree_spec_generate(shard=...)`` on this object must not leak into loops that
        prefill a full, unsharded cache."""
        for layer in self.model.layers:
            layer.self_attn.shard = None
        self.glide.cross_attn.shard = None

    @torch.inference_mode()
    def vanilla_generate(self, input_ids, prompt_length, max_gen_len=64, eos_id=151645):       # :552-585
        assert input_ids is not None, "please give the input"
        self._clear_shard()
        bsz = input_ids.size(0)
        output_ids = input_ids.new_zeros((bsz, max_gen_len))
        self.set_max_gen_len(max_gen_len)
        cache_lens = input_ids.new_zeros((bsz)).int()
        P = int(input_ids.size(1))
        self._set_hints(P, P)
        hidden_states = self.model.forward(input_ids, exec_type="prefill").last_hidden_state
        input_len = prompt_length
        rows = torch.arange(bsz, device=input_ids.device)
        output_ids[:, 0] = self.lm_head(hidden_states[rows, input_len - 1, :]).argmax(dim=-1)
        cache_lens += input_len.int()
        num = 0
        eos = self._stop_id(eos_id, "vanilla")
        _sync(input_ids)
        start_time = time.time()
        on_gpu = input_ids.is_cuda
        marks = []                      # per-step completion marks: the time of steps decoded past the first EOS is not counted
        if on_gpu:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        vs = self.begin_vanilla_decode(output_ids, cache_lens, input_len.int(), P)
        for step in range(1, max_gen_len):
            self.vanilla_step(vs)
            num += bsz
            if on_gpu:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
            else:
                marks.append(time.time() - start_time)
            # the reference reads the EOS test back after every token (:578); here every 16th, and the result is then cut
            # back to what the reference would have returned -- tokens, `num` AND elapsed time
            if eos is not None and (step % 16 == 0 or step == max_gen_len - 1):
                if bool((output_ids[:, :step + 1].eq(eos)).any()):
                    break
        _sync(input_ids)
        elapsed_time = time.time() - start_time
        if eos is not None:
            output_ids, num, stop = _truncate_after_eos_vanilla(output_ids, num, eos, bsz)
            if stop is not None and stop < len(marks):
                elapsed_time = ev0.elapsed_time(marks[stop - 1]) * 1e-3 if on_gpu else marks[stop - 1]
        return output_ids, num, elapsed_time

    # ------------------------------------------------------------------------------------------
    def begin_vanilla_decode(self, output_ids, cache_lens, input_len, prompt_bound: int):
        """State of the vanilla loop (``llama_glide.py:566-583``): ``output_ids`` [bsz, max_gen] with the first
        token in place, ``cache_lens`` [bsz] int32 valid rows of every cache, ``input_len`` [bsz] int32."""
        from types import SimpleNamespace
        dev = output_ids.device
        vs = SimpleNamespace(output_ids=output_ids, cache_lens=cache_lens, input_len=input_len, P=prompt_bound, step=0,
                             rows=torch.arange(output_ids.size(0), device=dev), graph=None, graph_stream=None,
                             graph_bound=0, graph_captures=0)
        vs.use_graphs = bool(dev.type == "cuda" and self.model.layers[-1].self_attn.shard is None and self.GRAPH_ROUNDS)
        return vs

    def _vanilla_device(self, vs):
        """One token, device work only: the step has no host-side parameter at all."""
        out, cl, il, rows = vs.output_ids, vs.cache_lens, vs.input_len, vs.rows
        cur = out[rows, (cl - il).long()].view(out.size(0), -1)
        hidden_states = self.model.forward(cur, cache_lens=cl, exec_type="decoding").last_hidden_state
        llm_output = self.ops.argmax_rows(self.lm_head(hidden_states[:, -1, :]))
        cl += 1
        out[rows, (cl - il).long()] = llm_output.view(-1)

    def vanilla_step(self, vs):
        """Decode one token.  On a GPU the step is captured into a HIP graph after GRAPH_AFTER eager steps and replayed."""
        vs.step += 1
        produced = False                                              # this call's token is already in output_ids
        if vs.use_graphs:
            try:
                if vs.graph is not None and vs.step + 1 > vs.graph_bound:     # (step s reads s rows beyond the prompt's)
                    vs.graph = None                                           # outgrown: capture the next tier's step
                if vs.graph is None and vs.step > self.GRAPH_AFTER:
                    vs.graph_bound = self._tier_bound(vs.step + 1, vs.output_ids.size(1))
                    self._set_hints(vs.P + vs.graph_bound, vs.P + vs.graph_bound)
                    cur = torch.cuda.current_stream()
                    if vs.graph_stream is None:
                        vs.graph_stream = torch.cuda.Stream()
                    vs.graph_stream.wait_stream(cur)
                    with torch.cuda.stream(vs.graph_stream):          # warm the capture stream's workspaces
                        self._vanilla_device(vs)
                    produced = True
                    cur.wait_stream(vs.graph_stream)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=vs.graph_stream):
                        self._vanilla_device(vs)
                    vs.graph = (graph, self.ops.workspace_tensors() if hasattr(self.ops, "workspace_tensors") else None)
                    vs.graph_captures += 1
                    return                                            # the warm-up step was this call's token
                if vs.graph is not None:
                    vs.graph[0].replay()
                    return
            except Exception as e:
                if torch.cuda.is_current_stream_capturing():
                    raise
                import warnings
                warnings.warn(f"HIP-graph capture of the vanilla step failed ({type(e).__name__}: {e}); running eagerly")
                vs.use_graphs = False
                vs.graph = None
                if produced:                                          # the warm-up step already produced this call's token
                    if vs.graph_stream is not None:
                        torch.cuda.current_stream().wait_stream(vs.graph_stream)
                    return
                # anything else (a failure in front of the warm-up step -- _set_hints, wait_stream -- or in a later
                # replay()) has not decoded this call's token: fall through to the eager step
        self._set_hints(vs.P + vs.step, vs.P + vs.step)
        self._vanilla_device(vs)

    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def spec_generate(self, input_ids, prompt_length, gamma=4, max_gen_len=64, eos_id=151645, temperature=0.0):   # :621-774
        return self._chain_generate(input_ids, prompt_length, gamma, max_gen_len, eos_id, temperature, drafter="glide")

    @torch.inference_mode()
    def magicdec_generate(self, input_ids, prompt_length, gamma=4, max_gen_len=64, eos_id=151645, temperature=0.0):  # :776-913
        """The MagicDec baseline of the reference's harness (``--method magicdec``): chain speculation where the
        drafter is the TARGET itself attending to a StreamingLLM cache (32 sink rows + the last 1024 prompt rows +
        what it generates), verified exactly like ``spec_generate``.  Same return tuple."""
        return self._chain_generate(input_ids, prompt_length, gamma, max_gen_len, eos_id, temperature, drafter="magicdec")

    @torch.inference_mode()
    def vanilla_torch_generate(self, input_ids, prompt_length, max_gen_len=64, eos_id=151645):                      # :587-619
        """``--method vanilla_torch``: the reference's autoregressive loop over its dense-PyTorch attention twins
        (``prefill_torch`` / ``decoding_torch``, llama.py:132-197).  There is one attention implementation here,
        so this is ``vanilla_generate`` (the twins compute the same function; they exist in the reference to
        time flash-attn against eager attention)."""
        return self.vanilla_generate(input_ids, prompt_length, max_gen_len=max_gen_len, eos_id=eos_id)

    def _chain_generate(self, input_ids, prompt_length, gamma, max_gen_len, eos_id, temperature, drafter):
        assert input_ids is not None, "please give the input"
        magic = drafter == "magicdec"       # (temperature > 0: the same rejection block in both loops, :715-736 = :854-875)
        self._clear_shard()
        bsz = input_ids.size(0)
        assert bsz == 1, "the reference's hot path is batch 1 (SURVEY section 1)"
        dev = input_ids.device
        self.set_max_gen_len(max_gen_len + 128)
        if not magic:
            self.glide.set_max_gen_len(max_gen_len + 128)
        P = int(input_ids.size(1))
        self._set_hints(P, P)
        hidden_states = self.model.forward(input_ids, exec_type="magicdec_prefill" if magic else "prefill").last_hidden_state
        input_len = prompt_length
        rows = torch.arange(bsz, device=dev)
        logits = self.lm_head(hidden_states[rows, input_len - 1, :])
        cache_lens = input_ids.new_zeros((bsz)).int() + input_len.int()
        if not magic:                                    # glide prefill
            hidden_states = self.model.embed_tokens(input_ids)
            position_ids = torch.arange(0, input_ids.size(1), device=dev)[None, :]
            position_embeddings = self.model.rotary_emb(hidden_states, position_ids)
            self.glide(hidden_states=hidden_states, position_embeddings=position_embeddings, llm_kv=self._last_kv(),
                       cache_lens=cache_lens.clone(), llm_kv_len=cache_lens.clone(), exec_type="prefill")
        st = self.begin_chain_decode(logits.argmax(dim=-1), cache_lens, input_len, P, gamma, max_gen_len, eos_id, temperature,
                                     drafter, first_logits=logits)
        _sync(input_ids)
        start_time = time.time()
        for out_index in range(1, max_gen_len):
            if not self.chain_round(st):
                break
        _sync(input_ids)
        elapsed_time = time.time() - start_time
        return st.output_ids, st.count, st.num, elapsed_time, st.spec_mask

    def begin_chain_decode(self, first_token, cache_lens, input_len, prompt_bound: int, gamma=4, max_gen_len=64, eos_id=151645,
                           temperature=0.0, drafter="glide", first_logits=None):
        """State of the chain-speculation loop right after the prefills (``llama_glide.py:641-668``): ``first_token`` [bsz]
        = the target's first generated token, ``cache_lens`` [bsz] int32 = valid rows of every cache (the draft's included),
        ``input_len`` [bsz] = prompt length, ``prompt_bound`` = host-side bound of it.  Also the entry point of benchmarks
        that time ``chain_round`` on synthetic KV (bench.py --method seq)."""
        from types import SimpleNamespace
        bsz = first_token.size(0)
        dev = first_token.device
        st = SimpleNamespace(gamma=gamma, magic=drafter == "magicdec", temperature=temperature, bsz=bsz, P=int(prompt_bound),
                             max_gen_len=max_gen_len, count=0, num=0, emitted=1, double_flag=False)
        st.output_ids = first_token.new_zeros((bsz, max_gen_len + gamma))
        st.spec_mask = first_token.new_zeros((bsz, max_gen_len + gamma))
        st.output_ids[:, 0] = first_token
        st.cache_lens = cache_lens
        st.draft_cache_lens = cache_lens.clone()
        st.input_len = input_len
        st.input_len_i32 = input_len.to(device=dev, dtype=torch.int32).view(bsz).contiguous()
        st.spec_buffer = st.output_ids.new_zeros((bsz, gamma + 1))
        st.spec_buffer[:, 0] = st.output_ids[:, 0]
        # temperature > 0 (:639-640,704,709): the draft's fp32 logits of every step, for the rejection test of :716-736
        st.spec_logits = None
        if temperature > 0:
            assert first_logits is not None
            st.spec_logits = first_logits.new_zeros((bsz, gamma + 1, first_logits.size(-1)), dtype=torch.float32)
            st.spec_logits[:, 0] = first_logits
        st.next_spec_start_token = st.output_ids.new_zeros((bsz, 2))
        st.next_spec_start_token[:, 0] = st.output_ids[:, 0]
        st.eos = self._stop_id(eos_id, "spec")
        st.stream_rows = self.model.layers[0].self_attn.STREAM_SINK + self.model.layers[0].self_attn.STREAM_WINDOW
        return st

    def chain_round(self, st) -> bool:
        """One round of ``spec_generate`` / ``magicdec_generate`` (``llama_glide.py:670-770``): gamma draft steps, one
        (gamma + 1)-row target pass, acceptance by cumulative match.  False = stop (EOS or the buffer is full)."""
        gamma, magic, bsz, P = st.gamma, st.magic, st.bsz, st.P
        dev = st.output_ids.device
        output_ids, cache_lens, draft_cache_lens = st.output_ids, st.cache_lens, st.draft_cache_lens
        spec_buffer, spec_logits, next_spec_start_token = st.spec_buffer, st.spec_logits, st.next_spec_start_token
        bound = P + st.emitted + gamma + 2
        self._set_hints(bound, bound)
        for spec_steps in range(0, gamma):
            if spec_steps == 0:
                if st.double_flag:
                    draft_ids = next_spec_start_token[:, 0:2]
                    position_ids = torch.arange(0, 2, device=dev)[None, :] + draft_cache_lens[:, None]
                else:
                    draft_ids = next_spec_start_token[:, 0, None]
                    position_ids = draft_cache_lens[:, None]
            else:
                draft_ids = spec_buffer[:, spec_steps, None]
                position_ids = draft_cache_lens[:, None]
            prenormed = None
            if magic:
                hidden_states = self.model.embed_tokens(draft_ids)
                position_embeddings = self.model.rotary_emb(hidden_states, position_ids)
            else:
                hidden_states, prenormed, position_embeddings = self._draft_head(draft_ids, position_ids=position_ids)
            if magic:        # the target drafts for itself over its streaming cache (:830-836)
                stream_lens = (draft_cache_lens - st.input_len.int() + st.stream_rows).to(torch.int32)
                self.model.set_kv_len_hint(st.stream_rows + st.emitted + gamma + 2)
                hidden_states = self.model.forward(draft_ids, position_embeddings=position_embeddings, cache_lens=stream_lens,
                                                   exec_type="magicdec_decoding").last_hidden_state
                self.model.set_kv_len_hint(bound)
            else:
                hidden_states = self.glide(hidden_states=hidden_states, position_embeddings=position_embeddings,
                                           llm_kv=self._last_kv(), cache_lens=draft_cache_lens,
                                           llm_kv_len=cache_lens, exec_type="decoding", prenormed=prenormed)
            if st.double_flag and spec_steps == 0:
                draft_cache_lens += 2                    # 1 + double_input (batch 1: the host knows the flag)
                current_logp = self.lm_head(hidden_states[:, -2:, :])
                spec_buffer[:, spec_steps + 1] = self.ops.argmax_rows(current_logp)[:, 1]
                if spec_logits is not None:
                    spec_logits[:, spec_steps + 1, :] = current_logp[:, 1, :]
            else:
                draft_cache_lens += 1
                current_logp = self.lm_head(hidden_states[:, -1, :])
                spec_buffer[:, spec_steps + 1] = self.ops.argmax_rows(current_logp).view(-1,)
                if spec_logits is not None:
                    spec_logits[:, spec_steps + 1, :] = current_logp
        hidden_states = self.model.forward(spec_buffer, cache_lens=cache_lens, exec_type="decoding").last_hidden_state
        llm_verify_logits = self.lm_head(hidden_states[:, -gamma - 1:, :])
        llm_verify_output = self.ops.argmax_rows(llm_verify_logits)
        accept = None
        if st.temperature > 0:                               # :715-736
            llm_verify_output, accept = self.ops.chain_accept_stochastic(spec_logits, llm_verify_logits, spec_buffer,
                                                                         llm_verify_output)
        # acceptance by cumulative match, verified ids + bonus token -> output_ids, cache_lens += correct_len, the next
        # round's start tokens and draft_cache_lens = cache_lens - double_input (:738-770): one launch, one host read
        state = self.ops.chain_commit(llm_verify_output, spec_buffer, output_ids, cache_lens, draft_cache_lens, st.input_len_i32,
                                      next_spec_start_token, st.eos, **({"accept_mask": accept} if accept is not None else {})).tolist()
        n_ok, hit = state[0][0], any(row[1] for row in state)
        st.double_flag = n_ok == gamma + 1
        st.count += n_ok - 1
        st.num += bsz
        st.emitted += n_ok
        if st.emitted - 1 + gamma + 2 > output_ids.size(1):
            return False
        return not hit

    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def tree_spec_generate(self, input_ids, prompt_length, tree_shape: Optional[List[int]] = None, max_gen_len=64,
                           eos_id=151645, temperature=0.0, shard=None):                       # :915-1126
        """``shard`` (``dist.KVShard``, every rank of the group calls with the same arguments): the prompt is prefilled
        and its KV kept sequence-sharded over the ranks (``_sharded_prefill``); decoding then runs replicated with one
        exchange per attention call.  Returns the same values on every rank."""
        assert input_ids is not None, "please give the input"
        bsz = input_ids.size(0)
        assert bsz == 1, "the reference's hot path is batch 1 (SURVEY section 1)"
        dev = input_ids.device
        self.set_max_gen_len(max_gen_len + 256)
        self.glide.set_max_gen_len(max_gen_len + 256)
        P = int(input_ids.size(1))
        input_len = prompt_length
        rows = torch.arange(bsz, device=dev)
        lens = input_len.to(device=dev, dtype=torch.int32).view(bsz)
        position_ids = torch.arange(0, input_ids.size(1), device=dev)[None, :]
        for layer in self.model.layers:                    # a model object may be reused with and without a shard
            layer.self_attn.shard = shard
        self.glide.cross_attn.shard = shard
        if shard is not None:
            if dev.type == "cuda" and not shard.peer_tried:
                # a mailbox slot holds the record of the widest pass: 128 rows x heads x (128 + 1) floats
                heads = max(self.config.num_attention_heads, self.glide.config.num_attention_heads)
                shard.enable_peer_exchange(bsz * 128 * heads * 129, dev)
            first = self._sharded_prefill(input_ids, input_len, position_ids, shard)
        else:
            # prefill LLM (:954-960)
            self._set_hints(P, P)
            hidden_states = self.model.forward(input_ids, exec_type="prefill").last_hidden_state
            first = self.lm_head(hidden_states[rows, input_len - 1, ...]).argmax(dim=-1)
            # prefill glide (:968-975)
            hidden_states = self.model.embed_tokens(input_ids)
            position_embeddings = self.model.rotary_emb(hidden_states, position_ids)
            self.glide(hidden_states=hidden_states, position_embeddings=position_embeddings, llm_kv=self._last_kv(),
                       cache_lens=lens.clone(), llm_kv_len=lens.clone(), exec_type="prefill")
        st = self.begin_tree_decode(first, lens, P, tree_shape, max_gen_len, eos_id, temperature=temperature)
        _sync(input_ids)
        start_time = time.time()
        for out_index in range(1, max_gen_len):
            if not (self.tree_round_stochastic(st) if temperature > 0 else self.tree_round(st)):
                break
            if shard is not None and out_index % 32 == 0:
                shard.raise_if_exchange_failed()
        _sync(input_ids)
        elapsed_time = time.time() - start_time
        if shard is not None:
            # a wait that gave up latches the exchange's error flag and every later wait returns at once: the rounds since
            # then merged stale mailbox records.  Never hand such tokens back (the peers fail the same way: a rank that
            # raises stops pushing, and their next wait times out)
            shard.raise_if_exchange_failed()
        return st.output_ids, st.count, st.num, elapsed_time, st.spec_mask

    def _sharded_prefill(self, input_ids, input_len, position_ids, shard):
        """Sequence-sharded prefill: rank r runs the target model on prompt rows [r*Ls, (r+1)*Ls) only
        (one all-gather of the layer's K/V rows per layer, ``LlamaAttention.sharded_prefill``) and keeps that slice of the
        KV; the tail rank computes the first token and broadcasts it.  The draft layer's cache is filled on every rank
        (its rows depend on the token ids only).  Work per rank: 1/W of the projections / MLP, between 1/W^2 (rank 0)
        and (2W-1)/W^2 (tail) of the attention."""
        bsz, P = input_ids.shape
        assert bsz == 1 and int(input_len.view(-1)[0]) == P, "sharded prefill: batch 1, unpadded prompt"
        lo = shard.start
        hi = P if shard.is_tail else min(P, lo + shard.Ls)
        assert hi > lo and (shard.world - 1) * shard.Ls < P <= shard.world * shard.Ls, \
            "sharded prefill: shard_rows must be ceil(prompt / world)-like (every rank owns at least one row)"
        for layer in self.model.layers:
            layer.self_attn.shard = shard
        self.glide.cross_attn.shard = shard
        shard.prefill_ctx = (lo, hi - lo, P)
        try:
            self._set_hints(P, P)
            hidden_states = self.model.forward(input_ids[:, lo:hi], position_ids=position_ids[:, lo:hi],
                                               exec_type="prefill").last_hidden_state
        finally:
            shard.prefill_ctx = None
        first = torch.zeros((bsz,), dtype=torch.int64, device=input_ids.device)
        if shard.is_tail:
            first = self.lm_head(hidden_states[:, hi - lo - 1, :]).argmax(dim=-1)
        shard.broadcast_from_tail(first)
        hidden_states = self.model.embed_tokens(input_ids)
        self.glide.prefill_cache_only(hidden_states, self.model.rotary_emb(hidden_states, position_ids))
        return first

    def begin_tree_decode(self, first_token, cache_lens, prompt_bound: int, tree_shape=None, max_gen_len=64, eos_id=151645,
                          temperature=0.0):
        """State of the round loop right after the two prefills (``llama_glide.py:927-991``).
        ``first_token`` [bsz] = the target's first generated token, ``cache_lens`` [bsz] int32 = valid
        rows of every KV cache, ``prompt_bound`` = host-side bound of it.  Also the entry point of
        synthetic-KV benchmarks, which fill the caches themselves instead of prefilling."""
        from types import SimpleNamespace
        dev = first_token.device
        bsz = first_token.shape[0]
        cand = [4, 16, 16, 16, 16] if tree_shape is None else list(tree_shape)
        acc_n = [1]
        for c in cand:
            acc_n.append(acc_n[-1] + c)
        Fn = acc_n[-1]                       # tree nodes incl. the root
        gamma = len(cand)
        R = Fn - 1 + gamma + 1               # verification rows: [a accepted | F-1 tree | pads]
        st = SimpleNamespace(cand=cand, acc_n=acc_n, Fn=Fn, gamma=gamma, R=R, P=prompt_bound, dev=dev, bsz=bsz)
        st.output_ids = torch.full((bsz, max_gen_len), self._tree_output_fill(eos_id), dtype=torch.int64, device=dev)  # :937 (G8)
        st.spec_mask = torch.zeros((bsz, max_gen_len), dtype=torch.int64, device=dev)
        st.output_ids[:, 0] = first_token
        cache_lens = cache_lens.to(torch.int32)              # the tree operators advance the three lengths in place
        st.cache_lens = cache_lens.clone()
        st.target_cache_lens_for_draft = cache_lens.clone()
        st.draft_cache_lens = cache_lens.clone()
        st.count, st.num = 0, bsz
        st.all_spec = torch.zeros((bsz, Fn), dtype=torch.int64, device=dev)
        st.all_spec[:, 0] = first_token
        st.acc_pad = torch.zeros((bsz, gamma + 1), dtype=torch.int64, device=dev)     # accepted ids of the last round
        st.acc_pad[:, 0] = first_token
        st.acc_ids = st.acc_pad[:, :1]
        st.a = 1                              # host mirror of acc_num (G9: the pad id is outside the vocab)
        st.emitted = 1                        # tokens written to output_ids so far (host mirror of emitted_dev)
        st.emitted_dev = torch.ones((bsz,), dtype=torch.int32, device=dev)
        # HIP graphs of the round, one per `a`: on a GPU; under a KV shard only when its exchange is the peer-store one
        # (kernel launches only: dist.PeerExchange) -- callers that bracket kernels with events switch it off per round
        sh = self.model.layers[-1].self_attn.shard
        st.use_graphs = bool(dev.type == "cuda" and (sh is None or sh.graph_safe) and self.GRAPH_ROUNDS)
        st.graphs, st.graph_stream, st.graph_pool, st.graphs_forced = {}, None, None, False
        st.graph_bound = self._tier_bound(1, max_gen_len)      # emitted-token bound the captured rounds are sized for
        st.graph_tiers, st.graph_captures = 1, 0               # diagnostics (tools/e2e_fullsize.py, the soak test)
        st.tree_mask = torch.zeros((bsz, Fn, Fn), dtype=torch.int64, device=dev)
        st.tree_mask[:, :, 0] = 1
        st.history_logp_sum = torch.zeros((bsz, Fn), dtype=torch.float32, device=dev)
        st.eos = self._stop_id(eos_id, "tree")
        st.arange_g = torch.arange(gamma + 2, device=dev)[None, :]
        st.temperature = float(temperature)
        if temperature > 0:
            # stochastic verification (:1093-1102): the draft's log-probs of every tree node are kept (`spec_logits`, :964),
            # up to gamma + 2 tokens come back per round, rounds run launch by launch (a host read inside)
            st.spec_logits = None                      # allocated at the first lm_head call (vocabulary size)
            st.acc_pad = torch.zeros((bsz, gamma + 2), dtype=torch.int64, device=dev)
            st.acc_pad[:, 0] = first_token
            st.acc_ids = st.acc_pad[:, :1]
            st.use_graphs = False
            st.input_len = cache_lens.clone()
            st.d0_rows = 1                             # rows of the next draft step 0 = width of the last acc_ids
        return st

    def tree_round(self, st) -> bool:
        """One draft-then-verify round (``llama_glide.py:997-1121``): 1 + (gamma-1) draft passes growing
        the beam tree, one R-row target pass, accept/collapse.  Returns False when generation must stop.

        Everything between two host reads is device work whose only host-side parameter is ``a``, the number of
        tokens accepted by the previous round (1..gamma+1).  With ``st.use_graphs`` the round is captured once per
        value of ``a`` into a HIP graph and replayed: the ~370 launches of a round (most of them 4-20 us draft
        kernels the host cannot issue fast enough) become one."""
        a = st.a
        state = None
        if st.use_graphs and (st.graphs_forced or st.num >= self.GRAPH_AFTER * st.bsz):
            state = self._graph_round(st, a)
        if state is None:
            # host bounds: no cache holds more than P + emitted (+ this round's speculative rows) valid rows
            self._set_hints(st.P + st.emitted + st.R, st.P + st.emitted + st.Fn)
            state = self._round_device(st, a)
        state = state.tolist()                                        # the round's ONE host read
        a, hit = state[0][0], any(row[1] for row in state)
        st.acc_ids = st.acc_pad[:, :a]
        st.a = a
        st.emitted += a
        st.count += a - 1
        st.num += st.bsz
        if st.emitted + st.gamma + 2 > st.output_ids.size(1):         # :1118
            return False
        if hit:                                                       # :1120
            return False
        return True

    def _tier_bound(self, emitted: int, total: int) -> int:
        """Emitted-token bound of the graph tier that holds `emitted` (GRAPH_TIER; the whole budget when tiers are off)."""
        tier = self.GRAPH_TIER
        if not tier or tier <= 0:
            return total
        return min(total, (emitted // tier + 1) * tier)

    def _graph_hints(self, st):
        # grid bounds of the current tier, so that a captured round stays valid until the generation leaves it
        bound = getattr(st, "graph_bound", None) or st.output_ids.size(1)
        self._set_hints(st.P + bound + st.R, st.P + bound + st.Fn)

    def _graph_warm(self, st, a: int):
        """Run the round eagerly ON the capture stream (the operator layer's workspaces are per stream; lazy
        one-time work -- weight packing, kernel attributes -- must not fall into a capture)."""
        cur = torch.cuda.current_stream()
        if st.graph_stream is None:
            st.graph_stream = torch.cuda.Stream()
        self._graph_hints(st)
        st.graph_stream.wait_stream(cur)
        with torch.cuda.stream(st.graph_stream):
            state = self._round_device(st, a)
        cur.wait_stream(st.graph_stream)
        st.graphs[a] = "warm"
        return state

    def _graph_capture(self, st, a: int):
        self._graph_hints(st)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st.graph_stream, pool=st.graph_pool):
            state = self._round_device(st, a)
        if st.graph_pool is None:
            st.graph_pool = graph.pool()
        # the graph holds raw pointers into the per-stream workspaces of the operator layer
        keep = self.ops.workspace_tensors() if hasattr(self.ops, "workspace_tensors") else None
        st.graphs[a] = (graph, state, keep)
        st.graph_captures += 1
        return st.graphs[a]

    def _graph_round(self, st, a: int):
        """Replay the HIP graph of a round that starts from ``a`` accepted tokens; the first round with a new ``a``
        runs eagerly (warm-up), the second one is captured.  Returns the round's state tensor, or None when the
        round has to run eagerly."""
        if st.graphs is False:
            return None
        if st.emitted > st.graph_bound:          # the generation has outgrown the tier its graphs were sized for
            st.graph_bound = self._tier_bound(st.emitted, st.output_ids.size(1))
            st.graphs = {}
            st.graph_tiers += 1
        try:
            g = st.graphs.get(a)
            if g is None:
                return self._graph_warm(st, a)
            if g == "warm":
                g = self._graph_capture(st, a)
            g[0].replay()
            return g[1]
        except Exception as e:          # capture is an optimisation: an environment that cannot do it runs eagerly
            if torch.cuda.is_current_stream_capturing():
                raise
            import warnings
            warnings.warn(f"HIP-graph capture of the decode round failed ({type(e).__name__}: {e}); running eagerly")
            st.graphs = False
            return None

    def prepare_tree_graphs(self, st):
        """Capture the round graphs of every accepted-token count now instead of on first use (benchmarks: keeps the
        captures out of the timed region).  The warm-up rounds run on a snapshot of the decode state: they only
        scribble on cache rows beyond the valid lengths."""
        if not st.use_graphs or st.graphs is False:
            return
        st.graphs_forced = True
        if st.emitted > st.graph_bound:
            st.graph_bound = self._tier_bound(st.emitted, st.output_ids.size(1))
            st.graphs = {}
        names = ("cache_lens", "target_cache_lens_for_draft", "draft_cache_lens", "tree_mask", "all_spec", "history_logp_sum",
                 "acc_pad", "output_ids", "emitted_dev")
        try:
            for a in range(1, st.gamma + 2):
                if isinstance(st.graphs.get(a), tuple):
                    continue
                if st.graphs.get(a) is None:
                    snap = {n: getattr(st, n).clone() for n in names}
                    self._graph_warm(st, a)
                    for n in names:
                        getattr(st, n).copy_(snap[n])
                self._graph_capture(st, a)
        except Exception as e:
            if torch.cuda.is_current_stream_capturing():
                raise
            import warnings
            warnings.warn(f"HIP-graph capture of the decode round failed ({type(e).__name__}: {e}); running eagerly")
            st.graphs = False

    def _round_device(self, st, a: int):
        """The device work of one round (no host read inside): returns state [bsz, 2] = (acc_num, eos hit).
        The length tensors are passed without the reference's ``.clone()``: their in-place updates (inside
        ``ops.tree_grow`` / ``tree_verify_inputs`` / ``tree_commit``) are ordered behind the kernels that read them on
        the same stream."""
        cand, acc_n, Fn, gamma, R, bsz = st.cand, st.acc_n, st.Fn, st.gamma, st.R, st.bsz
        tree_mask, all_spec, history_logp_sum = st.tree_mask, st.all_spec, st.history_logp_sum
        ops = self.ops
        last_attn = self.model.layers[-1].self_attn
        acc_ids = st.acc_pad[:, :a]
        # ---- D0: the a accepted tokens through the draft layer (:1003-1027).  At temperature > 0 the reference feeds the
        # whole zero-padded acc_ids row (its width, not acc_num, :1003): the extra rows only write cache rows that are
        # overwritten before they are read, but the ROW COUNT enters the bottom-right alignment of the causal cross-attention
        # (row i sees llm_kv_len - sq + i keys), so it is reproduced.
        n0 = a if st.temperature == 0 else st.d0_rows
        hidden_states, prenormed, position_embeddings = self._draft_head(st.acc_pad[:, :n0], pos_base=st.draft_cache_lens)
        hidden_states = self.glide(hidden_states=hidden_states, position_embeddings=position_embeddings,
                                   llm_kv=self._last_kv(), cache_lens=st.draft_cache_lens,
                                   llm_kv_len=st.target_cache_lens_for_draft, exec_type="decoding", prenormed=prenormed)
        # log_softmax + top-k of the draft's next-token distribution (:1019-1020), fused on the fp16 logits.  Under a shard
        # with `vocab_parallel` every rank multiplies by its slice of the lm_head only (dist.KVShard.head_select)
        vsh = last_attn.shard if (last_attn.shard is not None and last_attn.shard.vocab_parallel and st.temperature == 0) else None
        vocab_size = self.lm_head.out_features
        if vsh is not None:
            topk_logp, pred_ids = vsh.head_select(self.lm_head, hidden_states[:, a - 1, :], ops, k=cand[0])
        else:
            logits = self.lm_head(hidden_states[:, a - 1, :]).view(bsz, 1, -1)
            if st.temperature > 0:                     # spec_logits[:, 0] = current_logp (:1025, G8: log-probs, not logits)
                if st.spec_logits is None:
                    st.spec_logits = torch.zeros((bsz, Fn, vocab_size), dtype=torch.float32, device=logits.device)
                st.spec_logits[:, 0] = logits[:, 0].float().log_softmax(dim=-1)
            topk_logp, pred_ids = ops.logprob_topk(logits, None, cand[0])
        # the root's children (:1021-1027): tree_mask rows + diagonal, all_spec, log-prob sums, and
        # `draft_cache_lens += a - 1` -- one launch, which also hands back the next pass's positions and packed mask
        position_ids, mask_bits = ops.tree_grow(tree_mask, all_spec, history_logp_sum, topk_logp, pred_ids, vocab_size, 0, 1,
                                                base=st.draft_cache_lens, base_add=a - 1)
        # ---- D1..: tree levels (:1029-1075)
        for ms in range(1, gamma):
            lo, mid = acc_n[ms - 1], acc_n[ms]
            hidden_states, prenormed, position_embeddings = self._draft_head(all_spec[:, lo:mid], position_ids=position_ids)   # p + depth (:1032)
            hidden_states = self.glide(hidden_states=hidden_states, position_embeddings=position_embeddings,
                                       llm_kv=self._last_kv(), cache_lens=st.draft_cache_lens,
                                       llm_kv_len=st.target_cache_lens_for_draft, exec_type="tree_decoding",
                                       tree_mask=tree_mask[:, lo:mid, :mid], tree_mask_bits=mask_bits, prenormed=prenormed)
            # log_softmax + cumulative log-prob + flat top-k over (node, token) (:1046-1064), one fused operator
            if vsh is not None:
                topk_logp_sum, topk_indices = vsh.head_select(self.lm_head, hidden_states, ops, k=cand[ms],
                                                              history=history_logp_sum[:, lo:mid])
            else:
                level_logits = self.lm_head(hidden_states)
                if st.temperature > 0:                 # spec_logits[:, lo:mid] = current_logp (:1074)
                    st.spec_logits[:, lo:mid] = level_logits.float().log_softmax(dim=-1)
                topk_logp_sum, topk_indices = ops.logprob_topk(level_logits, history_logp_sum[:, lo:mid], cand[ms])
            # father = index // vocab, token = index % vocab, mask row = father's row + diagonal (:1056-1075)
            position_ids, mask_bits = ops.tree_grow(tree_mask, all_spec, history_logp_sum, topk_logp_sum, topk_indices,
                                                    vocab_size, lo, mid, base=st.draft_cache_lens, want_next=ms + 1 < gamma)
        # ---- V: one R-row target pass (:1078-1091); `draft_cache_lens += 1` (:1076) rides in the input assembly
        veri_spec, position_ids, mask_bits = ops.tree_verify_inputs(acc_ids, a, all_spec, tree_mask, st.cache_lens, R,
                                                                    bump=st.draft_cache_lens, bump_add=1)
        hidden_states = self.model.forward(veri_spec, position_ids=position_ids, cache_lens=st.cache_lens,
                                           exec_type="tree_decoding", tree_mask_bits=mask_bits).last_hidden_state
        hidden_states = hidden_states[:, a - 1:a + Fn - 1]
        if st.temperature > 0:
            return self.lm_head(hidden_states)         # the stochastic branch continues in tree_round_stochastic
        if vsh is not None:
            all_llm_pred = vsh.head_select(self.lm_head, hidden_states, ops, argmax=True).view(bsz, -1)
        else:
            all_llm_pred = ops.argmax_rows(self.lm_head(hidden_states))
        # ---- A: accept / reject tree collapse + last-layer KV row move (:1104-1116); the accepted rows start at
        # cache_lens + a - 1 (:1104), the cache lengths themselves advance in the commit below
        sh = last_attn.shard
        if sh is None:
            kv_lens, kc, vc = st.cache_lens, last_attn.K_Cache, last_attn.V_Cache
        elif sh.is_tail:      # the accepted rows live in the tail owner's local cache
            kv_lens, kc, vc = sh.local_len(st.cache_lens), last_attn.K_Cache, last_attn.V_Cache
        else:
            kv_lens, kc, vc = st.cache_lens, None, None
        acc_pad, acc_num_t, double_input, _ = ops.tree_collapse(all_spec, all_llm_pred, tree_mask, kv_lens, acc_n[-2],
                                                                gamma + 1, kc, vc, cache_len_add=a - 1, out_acc_ids=st.acc_pad)
        # emitted tokens -> output_ids (at the device-side offset), the EOS test on the whole buffer as the reference
        # does (:1120, G8), the tree state reset and `cache_lens += a`, `target_cache_lens_for_draft += acc_num`
        # (:1104-1117): one launch; its result is the round's ONE host read for (acc_num, eos flag)
        return ops.tree_commit(acc_pad, acc_num_t, st.output_ids, 0, st.eos, tree_mask, all_spec, history_logp_sum,
                               target_lens=st.cache_lens, target_add=a, draft_kv_lens=st.target_cache_lens_for_draft,
                               emitted_dev=st.emitted_dev)

    def tree_round_stochastic(self, st) -> bool:
        """One round at temperature > 0 (``llama_glide.py:997-1102,1110-1121``).  The draft passes and the verification pass
        are those of ``tree_round``; acceptance is ``verify_stochastic``.  The bookkeeping follows the reference's T > 0
        branch AS IT IS, which differs from its T = 0 branch (documented): ``cache_lens`` advances by
        ``acc_num - 1`` only (no ``+= 1``), the accepted rows of the last layer's KV are NOT compacted, and the whole
        zero-padded ``acc_ids`` row is written to ``output_ids`` at ``cache_lens - input_len`` -- so later rounds overwrite
        part of what earlier rounds wrote.  Reproduced token for token (tests/golden/verify_stochastic.npz)."""
        a = st.a
        if a + st.Fn - 1 > st.R:
            # the reference fails here too: gamma + 2 accepted tokens do not fit its veri_spec buffer (:1081)
            raise RuntimeError(f"stochastic round: {a} accepted tokens + {st.Fn - 1} tree nodes exceed the {st.R}-row "
                               f"verification batch (the reference raises at llama_glide.py:1081 in the same state)")
        # host bound of the valid rows, per round as in the eager T = 0 path: every round reads acc_num on the host anyway, and
        # the lengths (cache_lens advances by a - 1 per round, the draft's by acc_num) never exceed P + the sum of the accepted
        # counts -- NOT the whole token budget, which sized a 1k-prompt / 20k-budget run for 21k-row launches from round 1
        # 
        bound = min(st.emitted, st.output_ids.size(1))
        self._set_hints(st.P + bound + st.R, st.P + bound + st.Fn)
        llm_logits = self._round_device(st, a)                               # [bsz, Fn, V]
        st.cache_lens += a - 1                                               # :1094
        acc_ids, acc_num = self.verify_stochastic(st.all_spec, st.tree_mask, llm_logits, st.spec_logits, st.temperature)
        W = acc_ids.size(-1)
        cols = (st.cache_lens - st.input_len).long().unsqueeze(1) + torch.arange(W, device=acc_ids.device)[None, :]
        st.output_ids[torch.arange(st.bsz, device=acc_ids.device)[:, None], cols] = acc_ids          # :1102
        st.target_cache_lens_for_draft += acc_num.to(torch.int32)            # :1110
        n = int(acc_num[0])
        st.emitted += int(acc_num.max()) if st.bsz > 1 else n                # host mirror: an upper bound of every length's growth
        st.count += n - 1
        st.num += st.bsz
        st.tree_mask.fill_(0)
        st.tree_mask[:, :, 0] = 1
        st.all_spec.fill_(0)
        st.all_spec[:, 0] = acc_ids[:, n - 1]
        st.history_logp_sum.zero_()
        st.acc_pad.zero_()
        st.acc_pad[:, :W] = acc_ids
        st.acc_ids = st.acc_pad[:, :n]
        st.a = n
        st.d0_rows = W
        if int((st.cache_lens + acc_num.to(torch.int32) - st.input_len).max()) + st.gamma + 2 > st.output_ids.size(1):   # :1118
            return False
        if st.eos is not None and bool(st.output_ids.eq(st.eos).any()):      # :1120
            return False
        return True

    def verify_stochastic(self, input_ids, tree_mask, p_llm, p_ssm, temperature):               # :1177-1245
        """Drop-in for ``LlamaGlide.verify_stochastic``: (acc_ids [bsz, depth + 2] zero padded, acc_num [bsz])."""
        return self.ops.verify_stochastic(input_ids, tree_mask, p_llm, p_ssm, temperature)

    # ------------------------------------------------------------------------------------------
    def tree_verification(self, input_ids, output_ids, tree_mask, cache_lens, non_leaf_len):      # :1128-1175
        """Drop-in for ``LlamaGlide.tree_verification``: returns (acc_ids [bsz, acc_max],
        acc_num [bsz], double_input [bsz] int) and moves the last layer's KV rows."""
        depth = int(tree_mask.sum(dim=-1).max())
        last_attn = self.model.layers[-1].self_attn
        acc_pad, acc_num, double_input, _ = self.ops.tree_collapse(input_ids, output_ids, tree_mask, cache_lens, non_leaf_len,
                                                                   depth, last_attn.K_Cache, last_attn.V_Cache)
        return acc_pad[:, :int(acc_num.max())], acc_num, double_input


def _truncate_after_eos_vanilla(output_ids, num, eos, bsz):
    """The reference tests for EOS after every token and stops (``llama_glide.py:578``); this
    loop tests every 16 tokens to avoid a device sync per token, then restores the reference's
    visible result: tokens after the first EOS are zero, ``num`` counts forward passes up to it.  Returns
    (output_ids, num, steps the reference would have run or None)."""
    hit = output_ids.eq(eos)
    if not bool(hit.any()):
        return output_ids, num, None
    # the reference's loop always runs its first step: an EOS as the very first token stops it after step 1 (:571-579)
    first = max(int(hit.float().argmax(dim=-1).min()), 1)
    output_ids[:, first + 1:] = 0
    return output_ids, first * bsz, first
