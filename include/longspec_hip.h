/*
 * longspec_hip.h -- C ABI of liblongspec_hip.so: MI355X (gfx950) kernels for the
 * draft-then-verify decode round of LongSpec.
 *
 * The reference (sail-sg/LongSpec) has no FFI layer of its own; its operator
 * seams on this path are Python callables.  Each entry point below
 * replaces one of those seams and cites it (paths relative to the reference
 * root).  Conventions: extern "C"; raw DEVICE pointers + explicit dims/strides
 * (in elements); caller-allocated outputs and workspace (size-query functions);
 * every launch is stream-ordered on `stream` (a hipStream_t passed as void*);
 * no host synchronisation, no allocation, no hidden global state except the
 * thread-local last-error string.  Return value: 0 = ok, negative = LS_ERR_*.
 * Sequence lengths live on the DEVICE (int32), as in the reference
 * (`cache_seqlens=cache_lens`); `kv_len_hint` is a host-side upper bound used
 * only to size the launch grid.
 *
 * dtype: LS_F16 (the reference hard-codes fp16, longspec/test/llama_glide.py:474)
 * or LS_BF16.  head_dim must be 128 (longspec/test/llama.py:95).
 */
#ifndef LONGSPEC_HIP_H
#define LONGSPEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LS_OK 0
#define LS_ERR_INVALID_ARG (-1)
#define LS_ERR_UNSUPPORTED (-2)
#define LS_ERR_WORKSPACE (-3)
#define LS_ERR_LAUNCH (-4)

#define LS_F16 0
#define LS_BF16 1

/* how the "new key block" (tree / appended tokens) is computed and merged */
#define LS_NEW_NONE 0   /* prefix only                                                         */
#define LS_NEW_FLASH 1  /* flash_attn_with_kvcache(k=,v=) semantics: ONE softmax over prefix+new */
#define LS_NEW_TARGET 2 /* LlamaAttention.tree_part_fwd + fp16 merge  (llama.py:387,394-421)    */
#define LS_NEW_DRAFT 3  /* triton_tree_attn._fwd_kernel + fp32 merge  (llama_glide.py:300-329)  */

typedef struct ls_attn_desc {
    /* tensors (device) */
    const void* q;      /* [b, sq, H, 128]                     q_stride_{b,s,h}                  */
    void* k_cache;      /* [b, S, Hkv, 128]  read; written when scatter_new != 0                  */
    void* v_cache;      /*                                      kc_stride_{b,s,h} (shared by K,V) */
    const void* k_new;  /* [b, n_new_rows, Hkv, 128] or NULL    kn_stride_{b,s,h}                 */
    const void* v_new;
    const int32_t* cache_seqlens; /* [b] valid prefix rows L (device)                             */
    const uint32_t* mask_bits;    /* [b, sq, mask_words] bit j of row r = new-block key j visible */
    void* out;          /* [b, sq, H, 128] dtype                out_stride_{b,s,h}                */
    float* lse;         /* [b, H, sq] fp32 or NULL: soft-max log-normaliser (NONE/FLASH); the tree   */
                        /* kernel's L in LS_NEW_DRAFT mode; not available in LS_NEW_TARGET mode     */
    void* ev_start;     /* optional hipEvent_t pair recorded on `stream` around the stage-1 kernel  */
    void* ev_stop;      /* (profiling: per-launch duration of the streaming kernel), or NULL        */
    /* dims */
    int32_t b, sq, H, Hkv;
    int32_t dtype;         /* LS_F16 / LS_BF16                                                    */
    int32_t new_mode;      /* LS_NEW_*                                                            */
    int32_t n_new;         /* keys in the new block (0 for LS_NEW_NONE)                           */
    int32_t n_new_cached;  /* leading new-block keys that already sit in the cache at L + j       */
    int32_t mask_words;    /* uint32 words per mask row (>= ceil(n_new/32))                       */
    int32_t scatter_new;   /* write k_new/v_new rows into the caches at L + n_new_cached + i      */
    int32_t causal;        /* prefix: bottom-right causal alignment (flash-attn semantics)        */
    int32_t window_left;   /* prefix: sliding window to the left, -1 = none                       */
    int32_t n_app;         /* keys counted as appended for the bottom-right alignment (sk = L+n_app) */
    int32_t prescale_q;    /* LS_NEW_TARGET: last layer multiplies q by the scale first (G1)      */
    int32_t kv_len_hint;   /* host upper bound of max(cache_seqlens) -- grid sizing only          */
    int32_t n_splits;      /* 0 = choose automatically                                            */
    float softmax_scale;
    int64_t q_stride_b, q_stride_s, q_stride_h;
    int64_t kc_stride_b, kc_stride_s, kc_stride_h;
    int64_t kn_stride_b, kn_stride_s, kn_stride_h;
    int64_t out_stride_b, out_stride_s, out_stride_h;
} ls_attn_desc;

/* ---- library ---------------------------------------------------------------- */
int ls_version(void);
const char* ls_last_error(void);

/* ---- attention (K1..K7) ------------------------------------------------------- */

/* Bytes of scratch an attention call with this descriptor needs (split-KV partials). */
size_t ls_attn_workspace_bytes(const ls_attn_desc* d);

/* Number of prefix partials stage 1 writes for this descriptor (splits x key-slices). */
int ls_attn_num_parts(const ls_attn_desc* d);

/* Name of the stage-1 kernel that serves this descriptor ("attn_verify_kernel", "attn_partial_ws_kernel",
 * "attn_partial_kernel"): what a profiler will show for the call -- diagnostics / benchmark labels only. */
const char* ls_attn_kernel_name(const ls_attn_desc* d);

/* One fused attention call = stage 1 (split-KV partials + new-block part) then
 * stage 2 (log-sum-exp combine + reference-order merge), both on `stream`.
 *  LS_NEW_NONE / LS_NEW_FLASH : flash_attn_with_kvcache as called at
 *      longspec/test/llama.py:324,385 and llama_glide.py:261,265,297,300
 *  LS_NEW_TARGET : LlamaAttention.tree_decoding, llama.py:385-387 + tree_part_fwd :394-421
 *      (prefix flash-decoding + KV scatter + tree-masked part + fp16 merge)   [K1+K2+K3]
 *  LS_NEW_DRAFT  : GlideAttention.tree_decoding self-attn branch, llama_glide.py:300-302
 *      + triton_tree_part_fwd :309-329 (window prefix + Triton tree kernel + fp32 merge) [K5+K6] */
int ls_attn_fwd(const ls_attn_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* Diagnostic: workgroups (key split x kv head) that had to REDO their split since the last reset because a soft-max numerator
 * left the fp16 range of the split's fixed reference (csrc/attn.hip: the reference is the maximum over the split's first 64
 * keys; flash-attn's running maximum, SURVEY App. C, has no such event).  Synchronises the device; not for the decode path.
 * Returns the count (and zeroes it when reset != 0), -1 on a HIP error. */
long ls_attn_redo_count(int reset);

/* Stage 1 only: writes partials into the workspace (multi-GPU path). */
int ls_attn_partial(const ls_attn_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* Combine this rank's prefix partials into ONE normalised partial for the exchange:
 * o32 [b,sq,H,128] fp32, lse [b,H,sq] fp32 (the tree part stays in the workspace). */
int ls_attn_reduce_local(const ls_attn_desc* d, void* workspace, size_t workspace_bytes,
                         float* o32, float* lse, void* stream);

/* Stage 2 with externally gathered prefix partials (fixed rank order => deterministic):
 * parts_o [n_parts][b,sq,H,128] fp32, parts_lse [n_parts][b,H,sq]; the new-block part is
 * taken from `workspace` (written by ls_attn_partial).  N-way generalisation of
 * llama.py:385-387,420. */
int ls_attn_finish(const ls_attn_desc* d, const float* parts_o, const float* parts_lse, int n_parts,
                   int64_t part_o_stride, int64_t part_lse_stride, /* elements between parts; 0 = dense */
                   void* workspace, size_t workspace_bytes, void* stream);

/* N-way log-sum-exp merge of normalised partials (no new block): out [b,sq,H,128] dtype
 * and/or o32 + lse.  Any of out/o32/lse may be NULL. */
int ls_lse_merge(const float* parts_o, const float* parts_lse, int n_parts, int b, int sq, int H,
                 int dtype, void* out, float* o32, float* lse, void* stream);

/* tree_mask int64 [b,M,N] (non-zero = visible; llama_glide.py:984,1082-1085) -> packed
 * uint32 [b,M,words] (bit j%32 of word j/32), zero padded. */
int ls_pack_tree_mask(const int64_t* tree_mask, int b, int M, int N, uint32_t* bits, int words, void* stream);

/* ---- skinny linear layers (weight-streaming GEMM, M <= 80 token rows) ---------------- */

#define LS_EPI_NONE 0     /* y = x W^T (+ bias)                                                     */
#define LS_EPI_SILU_MUL 1 /* y = silu(x Wg^T) * (x Wu^T): w[0] = ls_linear_pack_gate_up(Wg, Wu), n[0] = N */
#define LS_EPI_QKV_ROPE 2 /* q|k|v projection + apply_rotary_pos_emb on q and k (llama.py:375-378): segments 0 and 1
                           * are packed by ls_linear_pack_rope and rotated with rope_cos/rope_sin, segment 2 is plain */

typedef struct ls_linear_desc {
    const void* x;        /* [M, K] dtype, row stride ldx (elements)                                  */
    const void* w[3];     /* weight segments, each [n[i], K] PACKED by ls_linear_pack_weight          */
    const void* bias[3];  /* [n[i]] dtype or NULL                                                     */
    void* y;              /* [M, sum n[i]] dtype, row stride ldy (LS_EPI_SILU_MUL: [M, n[0]])         */
    void* ev_start;       /* optional hipEvent_t pair recorded on `stream` around the kernel, or NULL */
    void* ev_stop;
    int32_t M, K;         /* M <= 80 token rows; K a multiple of 64 (>= 128)                          */
    int32_t n[3];         /* rows of each weight segment (multiples of 128 when n_seg > 1)            */
    int32_t n_seg;        /* 1..3 segments sharing x: q|k|v in one launch                             */
    int32_t dtype;        /* LS_F16 / LS_BF16                                                         */
    int32_t epilogue;     /* LS_EPI_*                                                                 */
    int32_t n_splits;     /* split-K factor, 0 = automatic (a function of N, K only -- never of M)    */
    int64_t ldx, ldy;
    const void* rope_cos; /* LS_EPI_QKV_ROPE: cos / sin [M, 128] dtype of the rows' positions (ls_rope_cos_sin) */
    const void* rope_sin;
    const void* residual; /* LS_EPI_NONE, optional: [M, sum n[i]] dtype (row stride ldr) added to the ROUNDED projection, */
    int64_t ldr;          /* `residual + mlp(x)` of the decoder layers (llama_glide.py:466); NULL = none                  */
    /* LlamaRMSNorm (llama.py:36 / qwen2.py:73-90) folded into the projection that consumes it.  `norm_weight` != NULL: x is
     * the UN-normalised residual stream and the launch multiplies `norm_weight * dtype(x32 * rsqrt(mean(x32^2) + eps))`
     * instead -- bit-identical to ls_rmsnorm_fwd followed by the plain launch.  The rows' sums of squares come as
     * `ssq_parts` fp32 partials per row (one per 64 columns of x, column order), written by the launch that produced x:
     * `ssq_out` != NULL (LS_EPI_NONE, one segment, N % 64 == 0) makes this launch such a producer: [M, N / 64] fp32
     * partial sums of the squares of the values it stores (after bias and residual).                                  */
    const void* norm_weight; /* [K] dtype or NULL                                                                        */
    const float* ssq_in;     /* [M, ssq_parts] fp32                                                                      */
    float* ssq_out;          /* [M, N / 64] fp32 or NULL                                                                 */
    int32_t ssq_parts;
    float norm_eps;
} ls_linear_desc;

/* Weights are streamed in the MFMA A-operand layout: pack each nn.Linear.weight [N, K] (row-major,
 * contiguous) ONCE after loading the checkpoint.  Packed size = ceil(N/64)*64 * K elements; for the
 * 64-row slab g, 32-wide k-step s and 16-row tile t the 1 KB block sits at element offset
 * ((g*(K/32) + s)*4 + t)*512 and holds, for lane l, W[64g + 16t + l%16][32s + 8(l/16) .. +8] at offset
 * 8*l (rows >= N are zero). */
size_t ls_linear_packed_bytes(int N, int K);
int ls_linear_pack_weight(const void* weight, void* packed, int N, int K, int dtype, void* stream);

/* gate_proj and up_proj [N, K] of one MLP packed as ONE matrix of 2N rows (ls_linear_packed_bytes(2N, K))
 * whose 16-row tiles alternate gate, up, gate, up ...: the operand of LS_EPI_SILU_MUL.  N % 16 == 0. */
int ls_linear_pack_gate_up(const void* gate_weight, const void* up_weight, void* packed, int N, int K,
                           int dtype, void* stream);

/* A q_proj / k_proj weight [heads*128, K] for LS_EPI_QKV_ROPE: same block structure, but within every head the
 * 16-row tiles are stored in the order 0,4,1,5,2,6,3,7 so that rows d and d+64 -- a rotary pair -- are finished by
 * the same wave.  The output columns are the plain ones. */
int ls_linear_pack_rope(const void* weight, void* packed, int N, int K, int dtype, void* stream);

/* Bytes of workspace (slab counters + split-K partials).  The workspace must be ZERO-FILLED once
 * before its first use; every call leaves the counter region zero again, so one workspace can serve
 * all launches of a stream. */
size_t ls_linear_workspace_bytes(const ls_linear_desc* d);

/* The projections of a decode pass: q/k/v/o_proj (longspec/test/llama.py:361-363,390;
 * llama_glide.py:248-250,268,285-287,305), LlamaMLP / Qwen2MLP.forward (transformers; vendored
 * qwen2.py:218-230: down_proj(act_fn(gate_proj(x)) * up_proj(x))), lm_head (llama_glide.py:960,1019,
 * 1046,1091).  fp32 accumulation in a fixed k order that does not depend on M; every linear's output
 * is rounded to dtype where nn.Linear rounds it.  LS_ERR_UNSUPPORTED for M > 80 or K % 64 != 0 (those
 * are plain library GEMMs: prefill). */
int ls_linear_fwd(const ls_linear_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* Cache hint for the launch `d` describes (no counterpart in the reference; writes nothing): the same grid requests the
 * first `units` register sets (8 KB each) of every wave's weight stream with the default cache policy, so that they sit in
 * the L2 of the XCD the matching workgroup of ls_linear_fwd(d) runs on.  Issued while a latency-bound kernel (RMSNorm,
 * attention finish) occupies the stream, it takes the HBM ramp off the projection that follows. */
int ls_linear_prefetch(const ls_linear_desc* d, int units, void* workspace, size_t workspace_bytes, void* stream);

/* ---- RMSNorm / RoPE (K8, K9) ------------------------------------------------ */

/* LlamaRMSNorm.forward (transformers; imported at longspec/test/llama.py:36; vendored
 * longspec/test/qwen2.py:82-87): y = w * dtype(x32 * rsqrt(mean(x32^2)+eps)).
 * Optional fused residual add: if residual != NULL, x <- x + residual is formed first
 * (rounded to dtype, as `residual + hidden_states` llama.py:492) and written to sum_out. */
int ls_rmsnorm_fwd(const void* x, const void* residual, const void* weight, void* y, void* sum_out,
                   int rows, int hidden, float eps, int dtype, void* stream);

/* LlamaRotaryEmbedding.forward (transformers; vendored qwen2.py:163-178):
 * cos/sin [rows,128] dtype from int64 positions and fp32 inv_freq[64]. */
int ls_rope_cos_sin(const int64_t* positions, const float* inv_freq, float attention_scaling,
                    void* cos, void* sin, int rows, int dtype, void* stream);

/* apply_rotary_pos_emb(q,k,cos,sin,unsqueeze_dim=2) (llama.py:378, llama_glide.py:294), in
 * place on q [rows,Hq,128] and k [rows,Hk,128] (row strides in elements). */
int ls_rope_apply(void* q, void* k, const void* cos, const void* sin, int rows, int Hq, int Hk,
                  int64_t q_row_stride, int64_t k_row_stride, int dtype, void* stream);

/* Tree positions: pos[r] = base[b] + sum_j mask[b,r,j] - 1 (llama.py:575-577, llama_glide.py:1032) */
int ls_tree_positions(const int64_t* tree_mask, const int32_t* base, int b, int M, int N,
                      int64_t* positions, void* stream);

/* ---- beam-tree growth / greedy verification on the lm_head logits ------------------------- */

/* Scratch for the two entry points below (per-chunk maxima, sums and candidates). */
size_t ls_topk_workspace_bytes(int rows, int vocab, int k);

/* `(lm_head(h).float().log_softmax(-1) + history[..., None]).view(-1).topk(k)` of one batch row
 * (longspec/test/llama_glide.py:1019-1020, 1046-1064): logits [rows, vocab] dtype (row stride ld), history
 * [rows] fp32 or NULL.  out_vals [k] fp32 descending, out_idx [k] int64 = row * vocab + column; equal
 * values are ordered by the smaller index.  vocab % 8 == 0, k <= 64, rows <= 128. */
int ls_logprob_topk(const void* logits, int rows, int vocab, int64_t ld, int dtype, const float* history, int k,
                    float* out_vals, int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream);

/* `lm_head(h).argmax(-1)` (llama_glide.py:578,1091): out_idx [rows] int64, first maximum of each row. */
int ls_argmax_rows(const void* logits, int rows, int vocab, int64_t ld, int dtype, int64_t* out_idx,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The same two operators with the lm_head SHARDED BY VOCABULARY over the ranks of a node (no counterpart in the reference, which
 * replicates: llama_glide.py:474).  A "record" is what stage 1 produces per (8192-logit chunk, row): the chunk's max, its sum of
 * exp(x - max) and its k largest logits with their GLOBAL columns -- rows * (2 + 2k) floats per chunk slot, chunk-major.
 *   ls_topk_stage1   this rank's slice logits_local [rows, vocab_local] (global columns col_base .. , col_base a multiple of
 *                    ls_topk_chunk()) -> records [nslots][rows][2 + 2k]; slots beyond the slice are written as empty records, so
 *                    that every rank can contribute the same number of slots to an all-gather;
 *   ls_topk_stage2   all ranks' slots in rank order = global chunk order -> the log-soft-max top-k (argmax = 0: out_vals [k],
 *                    out_idx [k] = row * vocab + column, history [rows] or NULL) or the per-row arg-max (argmax = 1: k = 1,
 *                    out_idx [rows], out_vals [rows] or NULL).
 * Bit-identical to ls_logprob_topk / ls_argmax_rows on the gathered logits for any number of ranks. */
int ls_topk_chunk(void);
int ls_topk_stage1(const void* logits_local, int rows, int vocab_local, int64_t ld, int dtype, int k, int col_base, int nslots,
                   float* records, void* stream);
int ls_topk_stage2(const float* records, int rows, int vocab, int k, int nslots, const float* history, int argmax, float* out_vals,
                   int64_t* out_idx, void* stream);

/* ---- beam-tree bookkeeping of a round (K10) ----------------------------------------------
 * The reference spells these steps as a few dozen tiny tensor ops per round
 * (longspec/test/llama_glide.py:1019-1121); each is one launch here, one workgroup per batch row. */

/* One more tree level (llama_glide.py:1021-1027 for the root's children, :1056-1075 deeper): new node
 * mid+j (j < k) is the child of node  lo + topk_idx[j] / vocab  and carries token  topk_idx[j] % vocab:
 *   all_spec[mid+j] = token;  logp_sum[mid+j] = topk_vals[j];
 *   tree_mask[mid+j, :] = tree_mask[father, :] + I[mid+j, :]          (the gather + diag_one of :1066-1067)
 * tree_mask [b,F,F] int64, all_spec [b,F] int64, logp_sum [b,F] fp32, topk_vals [b,k] fp32, topk_idx [b,k] int64.
 * positions [b,k] int64 / bits [b,k,words] (both or neither): what ls_tree_positions / ls_pack_tree_mask give
 * on tree_mask[mid:mid+k, :mid+k] for the draft pass over the new level, with base + base_add as the cache
 * length.  base [b] int32 (nullable) is advanced by base_add in place (`draft_cache_lens += acc - 1`, :1027). */
int ls_tree_grow(int64_t* tree_mask, int64_t* all_spec, float* logp_sum, const float* topk_vals,
                 const int64_t* topk_idx, int b, int F, int k, int64_t vocab, int lo, int mid,
                 int32_t* base, int base_add, int64_t* positions, uint32_t* bits, int words, void* stream);

/* Inputs of the verification pass (llama_glide.py:1078-1086).  R rows = [a accepted | F-1 tree nodes | pads]:
 *   veri_spec [b,R] int64 = [acc_ids[:a], all_spec[1:], 0...]        (acc_ids row stride acc_stride)
 *   mask = tril(ones(R,R)); mask[a:a+F-1, a:a+F-1] = tree_mask[1:,1:]; mask = tril(mask)
 *   positions [b,R] int64 = cache_lens + mask.sum(-1) - 1 (llama.py:575-577); bits [b,R,words] = mask != 0.
 * bump [b] int32 (nullable) += bump_add: the `draft_cache_lens += 1` of :1076. */
int ls_tree_verify_inputs(const int64_t* acc_ids, int64_t acc_stride, int a, const int64_t* all_spec,
                          const int64_t* tree_mask, int b, int F, int R, const int32_t* cache_lens,
                          int64_t* veri_spec, int64_t* positions, uint32_t* bits, int words,
                          int32_t* bump, int bump_add, void* stream);

/* LlamaGlide.tree_verification (longspec/test/llama_glide.py:1128-1175): all_spec/all_llm_pred [b,F]
 * int64, tree_mask [b,F,F] int64 (entries >= 0), cache_lens [b] int32 (+ cache_len_add = the `acc - 1`
 * of :1104).  Outputs: acc_ids [b,max_acc] int64 zero-padded, acc_num [b] int64, double_input [b] int32,
 * index_mapping [b,max_acc] int64 (-1 padded).  Moves the LAST target layer's KV rows
 * len+index_mapping[j] -> len+j (j < acc_num) in the same launch (k_cache NULL: no move). */
int ls_tree_collapse(const int64_t* all_spec, const int64_t* all_llm_pred, const int64_t* tree_mask,
                     const int32_t* cache_lens, int cache_len_add, int b, int F, int non_leaf_len, int max_acc,
                     int64_t* acc_ids, int64_t* acc_num, int32_t* double_input, int64_t* index_mapping,
                     void* k_cache, void* v_cache, int64_t kc_stride_b, int64_t kc_stride_s,
                     int row_elems, int dtype, void* stream);

/* LlamaGlide.verify_stochastic (longspec/test/llama_glide.py:1177-1245), temperature > 0: the speculative-sampling walk down
 * the draft tree, one workgroup per batch row.  all_spec [b,F] int64, tree_mask [b,F,F] int64, llm_logits [b,F,V] dtype
 * (target logits; strides in elements), spec_logp [b,Fs,V] fp32 (draft log-probs of the non-leaf nodes).  The reference
 * draws from Python's `random` and from torch.multinomial; both streams are handed in pre-drawn:
 *   mt_words [b,n_words] uint32 = the next raw outputs of Python's Mersenne Twister (random.getrandbits(32)): one word per
 *     random.choice attempt (>> (32 - bit_length(len))), two per random.random();  words_used [b] int32 = how many the walk
 *     consumed (-1: the buffer was too short) so that the host can re-synchronise its generator;
 *   exp_noise [b,V] dtype = torch.empty(V).exponential_(1): multinomial(p, 1) is argmax(p / noise) (ATen's one-sample path).
 * acc_ids [b,max_acc] int64 zero padded (max_acc >= tree depth + 2), acc_num [b] int64; workspace: b*V floats.
 * Kept from the reference on purpose: the acceptance ratio of child NODE s reads both distributions at vocabulary index s
 * (:1222); target probabilities are rounded to dtype after every operation (:1186,1231-1235). */
int ls_tree_verify_stochastic(const int64_t* all_spec, const int64_t* tree_mask, const void* llm_logits, int64_t logits_stride_b,
                              int64_t logits_stride_r, const float* spec_logp, int64_t logp_stride_b, int64_t logp_stride_r,
                              int b, int F, int Fs, int V, int dtype, float temperature, const uint32_t* mt_words, int n_words,
                              const void* exp_noise, int64_t* acc_ids, int max_acc, int64_t* acc_num, int32_t* words_used,
                              float* workspace, void* stream);

/* End of the round (llama_glide.py:1093-1121): output_ids[z, emitted + j] = acc_ids[z, j] (j < acc_num);
 * state[z] = (acc_num, any(output_ids[z, :out_cap] == eos)) for the round's single host read; the tree
 * state reset (tree_mask = 0, column 0 = 1; all_spec = 0, all_spec[0] = last accepted id; logp_sum = 0);
 * target_lens [b] int32 += target_add and draft_kv_lens [b] int32 += acc_num (both nullable).
 * emitted_dev [b] int32 (nullable): when given, the write offset is read from it instead of `emitted` and
 * advanced by acc_num -- a round replayed from a HIP graph carries no host-side integers. */
int ls_tree_commit(const int64_t* acc_ids, const int64_t* acc_num, int b, int max_acc, int64_t* output_ids,
                   int64_t out_stride, int out_cap, int emitted, int32_t* emitted_dev, int has_eos, int64_t eos,
                   int64_t* state, int64_t* tree_mask, int64_t* all_spec, float* logp_sum, int F,
                   int32_t* target_lens, int target_add, int32_t* draft_kv_lens, void* stream);

/* End of a chain-speculation round (spec_generate, llama_glide.py:738-770), gamma draft tokens:
 *   verification = cumprod(llm[:, :-1] == spec[:, 1:]); correct_len = sum + 1; llm[:, 1:] *= verification;
 *   output_ids[z, base+1 .. base+gamma] = llm[z, :gamma]; output_ids[z, base+correct_len] = bonus = llm[z, correct_len-1]
 *   (base = cache_lens - input_len); cache_lens += correct_len; draft_cache_lens = cache_lens - (correct_len == gamma+1);
 *   next_spec_start_token [b,2] = (llm[correct_len-2], llm[correct_len-1]) when every draft was accepted, else [0] = bonus;
 *   spec_buffer[z,0] = bonus; state [b,2] = (correct_len, any(output_ids[z, :base+correct_len+2] == eos)).
 * llm_verify_output / spec_buffer [b, gamma+1] int64 (both updated in place), lengths int32.
 * accept_mask [b, gamma] int64 or NULL: at temperature > 0 the verification is the cumulative product of the rejection-
 * sampling verdicts (:725-732) instead of the token comparison (:738). */
int ls_chain_commit(int64_t* llm_verify_output, int64_t* spec_buffer, int b, int gamma, int64_t* output_ids,
                    int64_t out_stride, int out_cap, int32_t* cache_lens, int32_t* draft_cache_lens,
                    const int32_t* input_len, int64_t* next_spec_start_token, int has_eos, int64_t eos,
                    int64_t* state, const int64_t* accept_mask, void* stream);

/* `embed_tokens(ids)` of a short pass (llama.py:579, llama_glide.py:1003,1030): out [n,hidden] =
 * table[ids] (table [vocab,hidden] dtype, ids int64 inside the vocabulary). */
int ls_embed_rows(const void* table, int64_t vocab, int hidden, int dtype, const int64_t* ids, int n,
                  void* out, void* stream);

/* Head of a decode pass in one launch: `embed_tokens(ids)` (llama.py:579, llama_glide.py:1003,1030) + the pass's RoPE table
 * (LlamaRotaryEmbedding.forward, llama.py:580 / llama_glide.py:1005-1006,1032-1033) + the first decoder layer's
 * `input_layernorm` (llama.py:487 via LlamaDecoderLayer.forward, llama_glide.py:437).  Bit-identical to
 * ls_embed_rows + ls_rope_cos_sin + ls_rmsnorm_fwd (the same device code); rows <= 128.
 *   ids [rows] int64; position of row i = positions[i] (int64) if given, else pos_base[i / q_len] + i % q_len + pos_add
 *   (int32 lengths on the device: `arange(q_len) + cache_lens[:, None]`);
 *   embeds [rows,hidden] = table[ids] (the residual stream), normed [rows,hidden] = RMSNorm(embeds) * norm_weight,
 *   cosv / sinv [rows,128]. */
int ls_pass_head(const void* table, int64_t vocab, int hidden, int dtype, const int64_t* ids, int rows,
                 const int64_t* positions, const int32_t* pos_base, int q_len, int pos_add, const float* inv_freq,
                 float attention_scaling, const void* norm_weight, float eps, void* embeds, void* normed, void* cosv,
                 void* sinv, void* stream);

/* ---- multi-GPU: peer exchange of the per-rank attention records -----------------
 * Replaces the one all-gather per attention call of a sequence-sharded prefix (the reference has no counterpart:
 * `device_map="auto"`, llama_glide.py:474) by peer stores into IPC-mapped mailboxes + flags, so that a decode round
 * containing it is pure kernel launches and replays from a HIP graph.  One object per process (= per GPU).
 *   create  -> handle (64 bytes, exchanged between the ranks by the host: torch.distributed, a pipe, ...)
 *           -> connect (all `world` handles in rank order; opens the peers' mailboxes)
 *           -> all_gather any number of times; every rank issues the same sequence of calls
 *   all_gather: gathered[r*stride_floats .. +n_floats) = rank r's `record` (n_floats <= cap_floats, multiples of 4,
 *   16-byte aligned).  Two launches on `stream`, no host synchronisation.  A peer that never delivers makes the
 *   wait give up after seconds and latches `timed_out` (ls_xchg_status; the gathered data is then invalid). */
#define LS_XCHG_MAX_WORLD 16
#define LS_XCHG_HANDLE_BYTES 64
typedef struct ls_xchg ls_xchg;
int ls_xchg_create(int rank, int world, size_t cap_floats, ls_xchg** out);
int ls_xchg_handle(ls_xchg* x, void* handle /* LS_XCHG_HANDLE_BYTES */);
int ls_xchg_connect(ls_xchg* x, const void* handles /* world * LS_XCHG_HANDLE_BYTES, rank order */);
int ls_xchg_all_gather(ls_xchg* x, const float* record, size_t n_floats, float* gathered, size_t stride_floats, void* stream);
int ls_xchg_status(ls_xchg* x, uint64_t* epoch, int* timed_out); /* synchronising; diagnostics only */
/* How long a wait polls before it gives up (default about 1 s; rank skew on a node is milliseconds).  Ranks that SHARE
 * one GPU (tests) stall each other for whole scheduling quanta and need tens of seconds.  Synchronising. */
int ls_xchg_set_timeout(ls_xchg* x, double seconds);
/* The same exchange fused into the two combine kernels of a sharded attention call: ls_attn_reduce_push =
 * ls_attn_reduce_local whose record goes straight into the peers' mailboxes (+ flags); ls_attn_finish_xchg =
 * ls_attn_finish over the mailbox slots of this exchange, read once every rank's flag is up.  One exchange =
 * one reduce_push followed by one finish_xchg on every rank (3 launches per attention call, 2 without a shard). */
int ls_attn_reduce_push(const ls_attn_desc* d, void* workspace, size_t workspace_bytes, ls_xchg* x, void* stream);
int ls_attn_finish_xchg(const ls_attn_desc* d, ls_xchg* x, void* workspace, size_t workspace_bytes, void* stream);
int ls_xchg_destroy(ls_xchg* x);

#ifdef __cplusplus
}
#endif
#endif /* LONGSPEC_HIP_H */
