"""The operator interface of ``longspec_amd.ops`` implemented with the CPU oracle, so that the
host logic (model wiring, cache-length state machine, tree growth, generate loops) can be
exercised without a GPU.  Lives under tests/: the product never imports it."""
import torch

from oracle import ref_ops


def rmsnorm(x, weight, eps, residual=None):
    if residual is not None:
        s = residual + x
        return ref_ops.rmsnorm(s, weight, eps), s
    return ref_ops.rmsnorm(x, weight, eps)


def rope_cos_sin(position_ids, inv_freq, attention_scaling, dtype):
    return ref_ops.rope_cos_sin(position_ids, inv_freq, attention_scaling, dtype)


def rope_apply_(q, k, cos, sin):
    q.copy_(ref_ops.apply_rope(q, cos, sin))
    if k.numel():
        k.copy_(ref_ops.apply_rope(k, cos, sin))
    return q, k


def pack_tree_mask(tree_mask):
    return tree_mask            # the oracle consumes the int mask directly


def tree_positions(tree_mask, base):
    pos = tree_mask.sum(dim=-1) - 1
    return pos + base[:, None].long() if base is not None else pos


def kvcache_attention(q, k_cache, v_cache, k=None, v=None, cache_seqlens=None, causal=False, window_size=(-1, -1),
                      return_softmax_lse=False, softmax_scale=None, kv_len_hint=None, n_splits=0):
    return ref_ops.kvcache_attention(q, k_cache, v_cache, k, v, cache_seqlens=cache_seqlens, causal=causal,
                                     window_size=window_size, return_softmax_lse=return_softmax_lse,
                                     softmax_scale=softmax_scale)


def verify_attention(q, k_new, v_new, k_cache, v_cache, cache_lens, mask_bits, last_layer, softmax_scale=1 / (128 ** 0.5),
                     kv_len_hint=None, n_splits=0):
    return ref_ops.target_verify_attention(q, k_new, v_new, k_cache, v_cache, cache_lens, mask_bits, last_layer, softmax_scale)


def draft_tree_attention(q, k_new, v_new, k_cache, v_cache, cache_lens, mask_bits, N, window=512, kv_len_hint=None):
    return ref_ops.draft_tree_self_attention(q, k_new, v_new, k_cache, v_cache, cache_lens, mask_bits, window)


def tree_grow(tree_mask, all_spec, logp_sum, topk_vals, topk_idx, vocab, lo, mid, base=None, base_add=0, want_next=True):
    """The reference's tensor ops for one more tree level (llama_glide.py:1021-1027 / :1056-1075), in place."""
    Fn = tree_mask.shape[1]
    k = topk_idx.shape[-1]
    hi = mid + k
    diag_one = torch.eye(Fn, dtype=torch.int64)[None].expand(tree_mask.shape[0], -1, -1)
    if lo == 0 and mid == 1:                                       # root's children (:1021-1024)
        tree_mask[:, 1:hi] += diag_one[:, 1:hi]
        all_spec[:, 1:hi] = topk_idx
    else:                                                          # :1056-1069
        father_ids = topk_idx // vocab + lo
        tree_mask[:, mid:hi] = torch.gather(tree_mask, 1, father_ids[:, :, None].expand(-1, -1, Fn)) + diag_one[:, mid:hi]
        all_spec[:, mid:hi] = topk_idx % vocab
    logp_sum[:, mid:hi] = topk_vals
    if base is not None and base_add:
        base += base_add
    if not want_next:
        return None, None
    ctm = tree_mask[:, mid:hi, :hi].contiguous()
    return tree_positions(ctm, base), pack_tree_mask(ctm)


def tree_verify_inputs(acc_ids, a, all_spec, tree_mask, cache_lens, R, bump=None, bump_add=0):
    """llama_glide.py:1078-1086 with the reference's tensor ops."""
    bsz, Fn = all_spec.shape
    veri_spec = tree_mask.new_zeros((bsz, R))
    veri_spec[:, :a] = acc_ids[:, :a]
    veri_spec[:, a:a + Fn - 1] = all_spec[:, 1:]
    new_tree_mask = torch.tril(torch.ones((R, R), dtype=torch.int64))[None].expand(bsz, -1, -1).contiguous()
    new_tree_mask[:, a:a + Fn - 1, a:a + Fn - 1] = tree_mask[:, 1:, 1:]
    new_tree_mask = torch.tril(new_tree_mask)
    if bump is not None and bump_add:
        bump += bump_add
    return veri_spec, tree_positions(new_tree_mask, cache_lens), pack_tree_mask(new_tree_mask)


def tree_commit(acc_ids, acc_num, output_ids, emitted, eos, tree_mask, all_spec, logp_sum, target_lens=None, target_add=0,
                draft_kv_lens=None, emitted_dev=None):
    """llama_glide.py:1093-1121 with the reference's tensor ops."""
    g = acc_ids.shape[1]
    if emitted_dev is not None:
        assert emitted_dev.numel() == 1, "the CPU restatement follows the reference: batch 1"
        emitted = int(emitted_dev[0])
        emitted_dev += acc_num.to(emitted_dev.dtype)
    sl = output_ids[:, emitted:emitted + g]
    keep = torch.arange(g)[None, :sl.shape[1]] < acc_num[:, None]
    sl.copy_(torch.where(keep, acc_ids[:, :sl.shape[1]], sl))
    hit = output_ids.eq(eos).any(dim=-1).to(torch.int64) if eos is not None else torch.zeros_like(acc_num)
    last = torch.gather(acc_ids, 1, (acc_num[:, None] - 1).clamp(min=0))[:, 0]
    tree_mask.fill_(0)
    tree_mask[:, :, 0] = 1
    all_spec.fill_(0)
    all_spec[:, 0] = last
    logp_sum.zero_()
    if target_lens is not None:
        target_lens += target_add
    if draft_kv_lens is not None:
        draft_kv_lens += acc_num.to(draft_kv_lens.dtype)
    return torch.stack([acc_num, hit], dim=1)


def chain_accept_stochastic(spec_logits, llm_verify_logits, spec_buffer, llm_verify_output):
    return ref_ops.chain_accept_stochastic(spec_logits, llm_verify_logits, spec_buffer, llm_verify_output)


def chain_commit(llm_verify_output, spec_buffer, output_ids, cache_lens, draft_cache_lens, input_len, next_spec_start_token, eos,
                 accept_mask=None):
    """llama_glide.py:738-770 with the reference's tensor ops (batch 1, as the reference); ``accept_mask`` = the T > 0
    branch's verification source (:732) instead of the token comparison (:738)."""
    bsz, g1 = llm_verify_output.shape
    gamma = g1 - 1
    rows = torch.arange(bsz)
    if accept_mask is not None:
        verification = accept_mask.to(torch.int64).cumprod(dim=-1)                         # :732
    else:
        verification = llm_verify_output[:, :-1].eq(spec_buffer[:, 1:]).cumprod(dim=-1)   # :738-740
    correct_len = verification.sum(dim=-1) + 1
    llm_verify_output[:, 1:] = llm_verify_output[:, 1:] * verification
    col = (cache_lens - input_len).long().unsqueeze(1) + torch.arange(1, gamma + 1)
    output_ids[rows.unsqueeze(1), col] = llm_verify_output[:, :gamma]
    bonus_token = llm_verify_output[rows, correct_len - 1]
    output_ids[rows, (cache_lens - input_len).long() + correct_len] = bonus_token
    base = (cache_lens - input_len).long()
    cache_lens += correct_len.int()
    double_input = correct_len.eq(gamma + 1).to(torch.int)
    for z in range(bsz):
        if int(double_input[z]):
            next_spec_start_token[z, 0] = llm_verify_output[z, correct_len[z] - 2]
            next_spec_start_token[z, 1] = llm_verify_output[z, correct_len[z] - 1]
        else:
            next_spec_start_token[z, 0] = bonus_token[z]
    spec_buffer[:, 0] = bonus_token
    draft_cache_lens.copy_(cache_lens - double_input)
    hit = torch.zeros_like(correct_len)
    if eos is not None:
        for z in range(bsz):
            hit[z] = int(output_ids[z, :int(base[z]) + int(correct_len[z]) + 2].eq(eos).any())
    return torch.stack([correct_len, hit], dim=1)


def embed_supported(ids, weight):
    return False


def tree_collapse(all_spec, all_llm_pred, tree_mask, cache_lens, non_leaf_len, max_acc, k_cache=None, v_cache=None,
                  cache_len_add=0, out_acc_ids=None):
    acc_ids, acc_num, dbl, imap = ref_ops.tree_verification(all_spec, all_llm_pred, tree_mask, non_leaf_len)
    if k_cache is not None:
        ref_ops.move_accepted_kv(k_cache, v_cache, cache_lens + cache_len_add, imap)
    b, n = acc_ids.shape
    pad_ids = torch.zeros((b, max_acc), dtype=torch.int64)
    pad_map = torch.full((b, max_acc), -1, dtype=torch.int64)
    for z in range(b):
        m = int(acc_num[z])
        pad_ids[z, :m] = acc_ids[z, :m]
        pad_map[z, :m] = imap[z, :m]
    if out_acc_ids is not None:
        out_acc_ids.copy_(pad_ids)
        pad_ids = out_acc_ids
    return pad_ids, acc_num, dbl, pad_map


def prefill_attention(q, k, v, k_cache, v_cache, window_left=-1, start=0, total=None):
    L = q.shape[1]
    k_cache[:, start:start + L] = k
    v_cache[:, start:start + L] = v
    if start == 0:
        return ref_ops.flash_attention(q, k, v, causal=True, window_size=(window_left, -1))
    # rows [start, start+L) of a longer prompt: bottom-right aligned causal attention over the rows so far
    lens = torch.full((q.shape[0],), start + L, dtype=torch.int32)
    return ref_ops.kvcache_attention(q, k_cache, v_cache, cache_seqlens=lens, causal=True, window_size=(window_left, -1))


class _ShardCall:
    """CPU counterpart of longspec_amd.ops.ShardedAttnCall (partial -> exchange -> finish)."""
    device = "cpu"

    def __init__(self, q, k_cache, v_cache, local_lens, causal=False, verify=None):
        self.q, self.kc, self.vc, self.lens, self.causal, self.verify = q, k_cache, v_cache, local_lens, causal, verify
        b, sq, H, D = q.shape
        self.n_o, self.n_lse = b * sq * H * D, b * H * sq
        self.record_floats = self.n_o + self.n_lse

    def partial(self, send):
        o, lse = ref_ops.kvcache_attention(self.q, self.kc, self.vc, cache_seqlens=self.lens, causal=self.causal,
                                           return_softmax_lse=True, keep_f32=True)
        send[:self.n_o] = o.reshape(-1)
        send[self.n_o:self.n_o + self.n_lse] = lse.reshape(-1)
        return send

    def finish(self, gathered):
        b, sq, H, D = self.q.shape
        W = gathered.shape[0]
        outs = []
        for z in range(b):
            parts_o = [gathered[w, :self.n_o].view(b, sq, H, D)[z] for w in range(W)]
            parts_l = [gathered[w, self.n_o:self.n_o + self.n_lse].view(b, H, sq)[z] for w in range(W)]
            outs.append(ref_ops.lse_merge(parts_o, parts_l))
        o32 = torch.stack([o for o, _ in outs], 0)
        lse = torch.stack([l for _, l in outs], 0)
        prefix_o = o32.to(self.q.dtype)
        if self.verify is None:
            return prefix_o
        k_new, v_new, mask, last_layer, scale = self.verify
        R = self.q.shape[1]
        for z in range(b):
            L = int(self.lens[z])
            self.kc[z, L:L + R] = k_new[z]
            self.vc[z, L:L + R] = v_new[z]
        cur, w = ref_ops.target_tree_part(self.q, k_new, v_new, mask, lse, last_layer, scale)
        one = torch.ones((), dtype=self.q.dtype)
        return prefix_o * w + cur * (one - w)


def sharded_verify_attention(q, k_new, v_new, k_cache, v_cache, local_lens, mask_bits, last_layer,
                             softmax_scale=1 / (128 ** 0.5), kv_len_hint=None, timing=None):
    return _ShardCall(q, k_cache, v_cache, local_lens, verify=(k_new, v_new, mask_bits, last_layer, softmax_scale))


def sharded_prefix_attention(q, k_cache, v_cache, local_lens, causal=False, kv_len_hint=None):
    return _ShardCall(q, k_cache, v_cache, local_lens, causal=causal)


def linear_supported(x, in_features):
    """The weight-streaming linear kernel is GPU-only: on the CPU the host logic uses torch's own F.linear."""
    return False


def logprob_topk(logits, history, k):
    """The reference's expression (llama_glide.py:1019-1020,1046-1064), verbatim."""
    if logits.dim() == 2:
        logits = logits.unsqueeze(1)
    lp = logits.float().log_softmax(dim=-1)
    if history is not None:
        lp = lp + history[:, :, None]
    return lp.view(lp.shape[0], -1).topk(dim=-1, k=k, largest=True, sorted=True)


def argmax_rows(logits):
    return logits.argmax(dim=-1)


def verify_stochastic(input_ids, tree_mask, p_llm, p_ssm, temperature):
    return ref_ops.verify_stochastic(input_ids, tree_mask, p_llm, p_ssm, temperature)
