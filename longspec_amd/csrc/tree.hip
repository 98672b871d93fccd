// Beam-tree bookkeeping of the draft-then-verify round: growing the tree by one level, assembling the
// verification pass's inputs, the accept/reject collapse, the end-of-round commit, and the token-embedding
// gather of the <= 80-row passes.  All of it is KB-scale integer work that the reference spells as a few dozen
// tiny tensor ops per round (llama_glide.py:1019-1121); here each step is ONE launch of one workgroup per batch
// row, every global load of a step issued before the first dependent use (these kernels are pure latency).
#include "ls_common.h"

namespace {

constexpr int MAXF = 1024;      // tree nodes
constexpr int TT = 1024;        // threads per workgroup (16 waves)

__device__ __forceinline__ long wave_sum_i64(long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- one more tree level (llama_glide.py:1021-1027 for the root's children, :1056-1075 deeper) ---------------
// wave w creates node mid + w (+16, ...): row copy of the father's mask row + the diagonal, the packed bits and the
// position of the new node for the draft pass that runs next.
__global__ __launch_bounds__(TT) void tree_grow_kernel(int64_t* __restrict__ tree_mask, int64_t* __restrict__ all_spec,
                                                       float* __restrict__ logp_sum, const float* __restrict__ vals,
                                                       const int64_t* __restrict__ idx, int F, int k, long vocab, int lo,
                                                       int mid, int32_t* base, int base_add, int64_t* __restrict__ positions,
                                                       uint32_t* __restrict__ bits, int words) {
    const int z = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t* mask = tree_mask + (long)z * F * F;
    const int hi = mid + k;
    const long b0 = base ? (long)base[z] + base_add : 0;
    for (int j = wave; j < k; j += TT / 64) {
        const long t = idx[(long)z * k + j];
        int father = (int)(t / vocab) + lo;
        father = father < 0 ? 0 : (father >= mid ? mid - 1 : father);       // memory safety only: fathers precede `mid` by construction
        const int row = mid + j;
        long s = 0;
        for (int c0 = 0; c0 < F; c0 += 64) {
            const int c = c0 + lane;
            long v = 0;
            if (c < F) {
                v = mask[(long)father * F + c] + (c == row ? 1 : 0);
                mask[(long)row * F + c] = v;
            }
            if (c < hi) s += v;
            const unsigned long long bal = __ballot(c < hi && v != 0);
            if (bits && lane == 0) {
                const int w = c0 >> 5;
                uint32_t* br = bits + ((long)z * k + j) * words;
                if (w < words) br[w] = (uint32_t)bal;
                if (w + 1 < words) br[w + 1] = (uint32_t)(bal >> 32);
            }
        }
        s = wave_sum_i64(s);
        if (lane == 0) {
            all_spec[(long)z * F + row] = t % vocab;
            logp_sum[(long)z * F + row] = vals[(long)z * k + j];
            if (positions) positions[(long)z * k + j] = b0 + s - 1;
        }
    }
    __syncthreads();                                   // every wave has read base[z]
    if (threadIdx.x == 0 && base && base_add) base[z] += base_add;
}

// ---- inputs of the verification pass (llama_glide.py:1078-1086) -----------------------------------------------
// R rows = [a accepted tokens | F-1 tree nodes | pads]; mask = tril(ones); mask[a:a+F-1, a:a+F-1] = tree_mask[1:,1:];
// mask = tril(mask).  Emits the token ids, the packed mask and the positions (mask.sum(-1) - 1 + cache_lens).
__global__ __launch_bounds__(TT) void tree_verify_inputs_kernel(const int64_t* __restrict__ acc_ids, long acc_stride, int a,
                                                                const int64_t* __restrict__ all_spec,
                                                                const int64_t* __restrict__ tree_mask, int F, int R,
                                                                const int32_t* __restrict__ cache_lens,
                                                                int64_t* __restrict__ veri_spec, int64_t* __restrict__ positions,
                                                                uint32_t* __restrict__ bits, int words, int32_t* bump,
                                                                int bump_add) {
    const int z = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t* mask = tree_mask + (long)z * F * F;
    const long L = cache_lens ? (long)cache_lens[z] : 0;
    const int t_end = a + F - 1;                       // first pad row
    for (int i = wave; i < R; i += TT / 64) {
        const bool tree_row = i >= a && i < t_end;
        long s = 0;
        for (int c0 = 0; c0 < R; c0 += 64) {
            const int c = c0 + lane;
            long v = 0;
            if (c < R && c <= i) v = (tree_row && c >= a) ? mask[(long)(i - a + 1) * F + (c - a + 1)] : 1;
            s += v;
            const unsigned long long bal = __ballot(v != 0);
            if (lane == 0) {
                const int w = c0 >> 5;
                uint32_t* br = bits + ((long)z * R + i) * words;
                if (w < words) br[w] = (uint32_t)bal;
                if (w + 1 < words) br[w + 1] = (uint32_t)(bal >> 32);
            }
        }
        s = wave_sum_i64(s);
        if (lane == 0) {
            positions[(long)z * R + i] = L + s - 1;
            veri_spec[(long)z * R + i] = i < a ? acc_ids[(long)z * acc_stride + i] : (i < t_end ? all_spec[(long)z * F + i - a + 1] : 0);
        }
    }
    if (threadIdx.x == 0 && bump && bump_add) bump[z] += bump_add;       // not read by this kernel
}

// ---- accept / reject tree collapse (LlamaGlide.tree_verification, llama_glide.py:1128-1175) --------------------
__global__ __launch_bounds__(TT) void tree_collapse_kernel(
    const int64_t* __restrict__ all_spec, const int64_t* __restrict__ all_pred, const int64_t* __restrict__ tree_mask,
    const int32_t* __restrict__ cache_lens, int len_add, int Fn, int non_leaf_len, int max_acc, int64_t* __restrict__ acc_ids,
    int64_t* __restrict__ acc_num, int32_t* __restrict__ double_input, int64_t* __restrict__ index_mapping, char* k_cache,
    char* v_cache, long kc_sb_bytes, long kc_ss_bytes, int row_bytes) {
    __shared__ unsigned long long best[MAXF];           // (mask*c) << 10 | (1023 - c): argmax with first-index ties
    __shared__ unsigned long long tsum[MAXF];
    __shared__ unsigned long long asum[MAXF];
    __shared__ int s_map[MAXF];
    __shared__ unsigned char verify[MAXF];
    __shared__ int s_last;
    __shared__ int s_count;
    const int z = blockIdx.x, tid = threadIdx.x;
    const int64_t* spec = all_spec + (long)z * Fn;
    const int64_t* pred = all_pred + (long)z * Fn;
    const int64_t* mask = tree_mask + (long)z * Fn * Fn;
    for (int r = tid; r < Fn; r += TT) {
        best[r] = 0;
        tsum[r] = 0;
        asum[r] = 0;
    }
    if (tid == 0) {
        s_last = 0;
        s_count = 0;
    }
    __syncthreads();
    // father[r] = argmax_c((mask - I)[r,c] * c): the largest mask*c over c != r, first c on ties, 0 if none (:1136);
    // tsum[r] = sum_c mask[r,c]
    const int total = Fn * Fn;
#pragma unroll 4
    for (int i = tid; i < total; i += TT) {
        const long m = mask[i];
        const int r = i / Fn, c = i - r * Fn;
        const long v = c == r ? (m - 1) * (long)c : m * (long)c;
        if (v > 0) atomicMax(&best[r], ((unsigned long long)v << 10) | (unsigned long long)(1023 - c));
        if (m != 0) atomicAdd(&tsum[r], (unsigned long long)m);
    }
    __syncthreads();
    for (int r = tid; r < Fn; r += TT) {
        const int father = best[r] ? 1023 - (int)(best[r] & 1023) : 0;
        verify[r] = (r == 0) || (pred[father] == spec[r]);                                   // :1138-1139
    }
    __syncthreads();
    // final[r] = sum_c mask[r,c]*verify[c] == sum_c mask[r,c]; last = argmax_r(final[r]*r)   (:1140-1144)
#pragma unroll 4
    for (int i = tid; i < total; i += TT) {
        const long m = mask[i];
        const int r = i / Fn, c = i - r * Fn;
        if (m != 0 && verify[c]) atomicAdd(&asum[r], (unsigned long long)m);
    }
    __syncthreads();
    for (int r = tid; r < Fn; r += TT)
        if (r > 0 && asum[r] == tsum[r]) atomicMax(&s_last, r);
    __syncthreads();
    const int last = s_last;
    // selected columns of mask[last] in ascending order (:1147-1154)
    if (tid < 64) {
        int base = 0;
        for (int c0 = 0; c0 < Fn; c0 += 64) {
            const int c = c0 + tid;
            const bool sel = (c < Fn) && (mask[(long)last * Fn + c] != 0);
            const unsigned long long bal = __ballot(sel);
            if (sel) s_map[base + __popcll(bal & ((1ull << tid) - 1ull))] = c;
            base += __popcll(bal);
        }
        if (tid == 0) s_count = base;
    }
    __syncthreads();
    const int n_acc = s_count;
    if (tid == 0) {
        acc_num[z] = n_acc;
        double_input[z] = last >= non_leaf_len ? 1 : 0;
    }
    for (int j = tid; j < max_acc; j += TT) {
        const int src = j < n_acc ? s_map[j] : -1;
        index_mapping[(long)z * max_acc + j] = src;
        acc_ids[(long)z * max_acc + j] = src >= 0 ? pred[src] : 0;       // :1155
    }
    // move the last target layer's KV rows L + map[j] -> L + j   (:1159-1173)
    if (k_cache == nullptr) return;
    const long L = (long)cache_lens[z] + len_add;
    const int chunks_per_row = row_bytes / 16;
    const int n_move = min(n_acc, max_acc);
    const int total_ch = n_move * chunks_per_row;
    constexpr int MAXC = 2;                                 // 16-byte chunks per thread and tensor in flight
    for (int c0 = 0; c0 < total_ch; c0 += TT * MAXC) {
        uint4 kb[MAXC], vb[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ix = c0 + i * TT + tid;
            if (ix < total_ch) {
                const int j = ix / chunks_per_row, ch = ix % chunks_per_row;
                const long src = (long)z * kc_sb_bytes + (L + s_map[j]) * kc_ss_bytes + (long)ch * 16;
                kb[i] = *reinterpret_cast<const uint4*>(k_cache + src);
                vb[i] = *reinterpret_cast<const uint4*>(v_cache + src);
            }
        }
        // rows move towards lower indices (map[j] >= j): a group may only be written once every read of it is done
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ix = c0 + i * TT + tid;
            if (ix < total_ch) {
                const int j = ix / chunks_per_row, ch = ix % chunks_per_row;
                const long dst = (long)z * kc_sb_bytes + (L + j) * kc_ss_bytes + (long)ch * 16;
                *reinterpret_cast<uint4*>(k_cache + dst) = kb[i];
                *reinterpret_cast<uint4*>(v_cache + dst) = vb[i];
            }
        }
        __syncthreads();
    }
}

// ---- end of the round (llama_glide.py:1093-1121) ----------------------------------------------------------------
// emitted tokens -> output_ids, the whole-buffer EOS test (:1120), the reset of the tree state for the next round
// (:1098-1102) and the length bookkeeping (:1104-1117), behind ONE host read of state[z] = (acc_num, eos hit).
__global__ __launch_bounds__(TT) void tree_commit_kernel(const int64_t* __restrict__ acc_ids, const int64_t* __restrict__ acc_num,
                                                         int max_acc, int64_t* __restrict__ output_ids, long out_stride,
                                                         int out_cap, int emitted_arg, int32_t* emitted_dev, int has_eos,
                                                         int64_t eos,
                                                         int64_t* __restrict__ state, int64_t* __restrict__ tree_mask,
                                                         int64_t* __restrict__ all_spec, float* __restrict__ logp_sum, int F,
                                                         int32_t* target_lens, int target_add, int32_t* draft_kv_lens) {
    __shared__ int s_hit;
    const int z = blockIdx.x, tid = threadIdx.x;
    const int n = (int)acc_num[z];
    const int emitted = emitted_dev ? emitted_dev[z] : emitted_arg;       // device-side counter: graph replays carry no host ints
    int64_t* out = output_ids + (long)z * out_stride;
    if (tid == 0) s_hit = 0;
    if (tid < max_acc && tid < n && emitted + tid < out_cap) out[emitted + tid] = acc_ids[(long)z * max_acc + tid];
    const int64_t next_root = n >= 1 ? acc_ids[(long)z * max_acc + min(n, max_acc) - 1] : 0;
    __syncthreads();
    if (has_eos) {
        int hit = 0;
        for (int i = tid; i < out_cap; i += TT) hit |= out[i] == eos;
        if (hit) atomicOr(&s_hit, 1);
    }
    int64_t* mask = tree_mask + (long)z * F * F;
    for (int i = tid; i < F * F; i += TT) mask[i] = (i % F) == 0 ? 1 : 0;
    for (int i = tid; i < F; i += TT) {
        all_spec[(long)z * F + i] = i == 0 ? next_root : 0;
        if (logp_sum) logp_sum[(long)z * F + i] = 0.f;
    }
    __syncthreads();
    if (tid == 0) {
        state[2 * z] = n;
        state[2 * z + 1] = s_hit;
        if (target_lens) target_lens[z] += target_add;
        if (draft_kv_lens) draft_kv_lens[z] += n;
        if (emitted_dev) emitted_dev[z] = emitted + n;
    }
}

// ---- end of a chain-speculation round (llama_glide.py:738-770) -------------------------------------------------------
// verification = cumprod(llm[:, :-1] == spec[:, 1:]); correct_len = sum + 1; the verified ids and the bonus token go to
// output_ids; cache_lens += correct_len; the next round's start tokens; state = (correct_len, eos hit).  One wave per row.
__global__ __launch_bounds__(64) void chain_commit_kernel(int64_t* __restrict__ llm, int64_t* __restrict__ spec, int G,
                                                          int64_t* __restrict__ output_ids, long out_stride, int out_cap,
                                                          int32_t* cache_lens, int32_t* draft_cache_lens,
                                                          const int32_t* __restrict__ input_len, int64_t* __restrict__ next_start,
                                                          int has_eos, int64_t eos, int64_t* __restrict__ state,
                                                          const int64_t* __restrict__ accept) {
    const int z = blockIdx.x, lane = threadIdx.x;
    int64_t* l = llm + (long)z * (G + 1);
    int64_t* sp = spec + (long)z * (G + 1);
    int64_t* out = output_ids + (long)z * out_stride;
    // lane i < G: does draft token i+1 match the target's prediction after token i?  correct = leading matches
    // (temperature > 0: the rejection-sampling verdicts of llama_glide.py:725-732 instead of the comparison)
    const bool match = lane < G ? (accept ? accept[(long)z * G + lane] != 0 : l[lane] == sp[lane + 1]) : false;
    const unsigned long long bal = __ballot(match);
    const int lead = __ffsll((long long)~bal) - 1;                      // first lane that does not match (>= 0: lane G never does)
    const int correct = lead + 1;                                       // 1 .. G+1
    const long base = (long)cache_lens[z] - (long)input_len[z];
    __syncthreads();
    if (lane >= 1 && lane <= G && lane - 1 >= lead) l[lane] = 0;        // llm[:, 1:] *= verification
    __syncthreads();
    if (lane >= 1 && lane <= G && base + lane < out_cap) out[base + lane] = l[lane - 1];
    const int64_t bonus = l[correct - 1];
    __syncthreads();
    if (lane == 0) {
        if (base + correct < out_cap) out[base + correct] = bonus;
        const int di = correct == G + 1 ? 1 : 0;
        if (di) {
            next_start[2 * z] = l[correct - 2];
            next_start[2 * z + 1] = l[correct - 1];
        } else {
            next_start[2 * z] = bonus;
        }
        sp[0] = bonus;
        const int cl = cache_lens[z] + correct;
        cache_lens[z] = cl;
        draft_cache_lens[z] = cl - di;
        state[2 * z] = correct;
    }
    __syncthreads();
    // EOS on output_ids[:, :emitted + 1] with emitted = base + 1 + correct (the host's counter after this round)
    int hit = 0;
    if (has_eos) {
        const long n = min((long)out_cap, base + correct + 2);
        for (long i = lane; i < n; i += 64) hit |= out[i] == eos;
    }
    const unsigned long long hb = __ballot(hit != 0);
    if (lane == 0) state[2 * z + 1] = hb != 0ull ? 1 : 0;
}

// ---- token embedding of a short pass: out[i,:] = table[ids[i],:] ----------------------------------------------------
__global__ __launch_bounds__(256) void embed_rows_kernel(const char* __restrict__ table, const int64_t* __restrict__ ids,
                                                         long vocab, int row_bytes, char* __restrict__ out) {
    long id = ids[blockIdx.x];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // memory safety; ids come from `% vocab` / argmax
    const char* src = table + id * (long)row_bytes;
    char* dst = out + (long)blockIdx.x * row_bytes;
    for (int o = threadIdx.x * 16; o < row_bytes; o += 256 * 16)
        *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(src + o);
}


// ---- temperature > 0: LlamaGlide.verify_stochastic (llama_glide.py:1177-1245) ---------------------------------------
// The speculative-sampling walk down the draft tree of ONE batch row per workgroup.  The reference draws from Python's
// `random` (random.choice over the remaining children, random.random) and from torch.multinomial; the host hands the
// kernel both streams pre-drawn -- `mt_words`: the next raw 32-bit outputs of the Mersenne Twister (random.choice(seq) =
// seq[getrandbits(k) rejection-sampled below len(seq)], getrandbits(k <= 32) = one word >> (32 - k); random.random() =
// ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53), `exp_noise`: the Exponential(1) row of multinomial's one-sample fast path
// (result = argmax(p / noise)) -- and reads back how many words were consumed.  Everything the reference keeps in the
// activation dtype is rounded to it here after every operation (soft-max, p + 1e-9, the residual max(p - q, 0) / sum).
// Reference quirk kept: the acceptance ratio of child node s reads both distributions at VOCABULARY index s (:1222).
template <typename E>
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < TT / 64; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

template <typename E>
__global__ __launch_bounds__(TT) void tree_verify_stochastic_kernel(
    const int64_t* __restrict__ all_spec, const int64_t* __restrict__ tree_mask, const typename E::T* __restrict__ logits,
    long lg_sb, long lg_sr, const float* __restrict__ logp, long lp_sb, long lp_sr, int F, int Fs, int V, float temperature,
    const uint32_t* __restrict__ mt_words, int n_words, const typename E::T* __restrict__ noise, int64_t* __restrict__ acc_ids,
    int max_acc, int64_t* __restrict__ acc_num, int32_t* __restrict__ words_used, float* __restrict__ prow_all) {
    __shared__ int father[MAXF];
    __shared__ int kids[MAXF];
    __shared__ float red[TT / 64];
    __shared__ int sh_n, sh_cur, sh_taken, sh_reject, sh_w, sh_bad;
    __shared__ float sh_best;
    __shared__ int sh_besti;
    __shared__ long path[16];
    const int z = blockIdx.x, tid = threadIdx.x;
    const int64_t* mask = tree_mask + (long)z * F * F;
    const typename E::T* lg = logits + (long)z * lg_sb;
    const float* lp = logp + (long)z * lp_sb;
    const uint32_t* words = mt_words + (long)z * n_words;
    float* prow = prow_all + (long)z * V;
    // fathers: the largest ancestor index other than the node itself (0 for the root)
    for (int u = tid; u < F; u += TT) {
        int f = 0;
        for (int j = 0; j < F; ++j)
            if (j != u && mask[(long)u * F + j] != 0) f = j;
        father[u] = f;
    }
    if (tid == 0) {
        sh_cur = 0;
        sh_w = 0;
        sh_bad = 0;
        path[0] = all_spec[(long)z * F];
    }
    __syncthreads();
    int n_path = 1;
    int row_of = -1;                          // node whose (possibly residual) target distribution prow holds
    float mq = 0.f, sq = 1.f;                 // soft-max statistics of the draft row of the current node
    auto load_rows = [&](int node) {          // prow = softmax(logits[node] / T) in E;  (mq, sq) of softmax(logp[node] / T) in fp32
        float m = -INFINITY;
        for (int j = tid; j < V; j += TT) m = fmaxf(m, round_to<E>(E::to_f32(lg[(long)node * lg_sr + j]) / temperature));
        m = block_reduce<E>(m, true, red);
        float sm = 0.f;
        for (int j = tid; j < V; j += TT) sm += expf(round_to<E>(E::to_f32(lg[(long)node * lg_sr + j]) / temperature) - m);
        sm = block_reduce<E>(sm, false, red);
        for (int j = tid; j < V; j += TT)
            prow[j] = round_to<E>(expf(round_to<E>(E::to_f32(lg[(long)node * lg_sr + j]) / temperature) - m) / sm);
        if (node < Fs) {
            float m2 = -INFINITY;
            for (int j = tid; j < V; j += TT) m2 = fmaxf(m2, lp[(long)node * lp_sr + j] / temperature);
            m2 = block_reduce<E>(m2, true, red);
            float s2 = 0.f;
            for (int j = tid; j < V; j += TT) s2 += expf(lp[(long)node * lp_sr + j] / temperature - m2);
            s2 = block_reduce<E>(s2, false, red);
            mq = m2;
            sq = s2;
        }
        __syncthreads();
    };
    auto qval = [&](int node, int j) -> float { return expf(lp[(long)node * lp_sr + j] / temperature - mq) / sq; };
    while (true) {
        const int cur = sh_cur;
        if (tid == 0) {                       // children of cur in increasing node order
            int n = 0;
            for (int u = 0; u < F; ++u)
                if (u != cur && father[u] == cur) kids[n++] = u;
            sh_n = n;
            sh_taken = -1;
        }
        __syncthreads();
        if (sh_n == 0) break;
        load_rows(cur);
        row_of = cur;
        while (true) {                        // draw children until one is accepted or none is left
            if (tid == 0) {
                int n = sh_n, w = sh_w;
                sh_reject = -1;
                if (n > 0) {
                    const int k = 32 - __clz(n);
                    unsigned r;
                    do {
                        if (w >= n_words) { sh_bad = 1; r = 0; break; }
                        r = words[w++] >> (32 - k);
                    } while (r >= (unsigned)n);
                    const int s = kids[r];
                    double rnd = 0.0;
                    if (w + 2 <= n_words) {
                        const unsigned a = words[w] >> 5, b2 = words[w + 1] >> 6;
                        w += 2;
                        rnd = ((double)a * 67108864.0 + (double)b2) * (1.0 / 9007199254740992.0);
                    } else sh_bad = 1;
                    const float pv = round_to<E>(prow[s] + 1e-9f);
                    const float qv = qval(cur, s) + 1e-9f;
                    const float ratio = pv / qv;
                    if ((float)rnd <= ratio) sh_taken = s;
                    else {
                        sh_reject = s;
                        int o = 0;
                        for (int i = 0; i < n; ++i)
                            if (kids[i] != s) kids[o++] = kids[i];
                        sh_n = o;
                    }
                    sh_w = w;
                }
            }
            __syncthreads();
            if (sh_taken >= 0 || sh_reject < 0) break;
            // residual distribution of the target at cur: max(p - q, 0), renormalised, every step rounded to E
            float part = 0.f;
            for (int j = tid; j < V; j += TT) {
                float v = round_to<E>(prow[j] - qval(cur, j));
                v = v > 0.f ? v : 0.f;
                prow[j] = v;
                part += v;
            }
            const float tot = round_to<E>(block_reduce<E>(part, false, red));
            if (tot > 0.f)
                for (int j = tid; j < V; j += TT) prow[j] = round_to<E>(prow[j] / tot);
            __syncthreads();
            if (sh_n == 0) break;
        }
        if (sh_taken < 0) break;
        if (tid == 0) {
            if (n_path < 15) path[n_path] = all_spec[(long)z * F + sh_taken];
            sh_cur = sh_taken;
        }
        ++n_path;
        __syncthreads();
    }
    const int cur = sh_cur;
    if (row_of != cur) load_rows(cur);
    // torch.multinomial(p, 1): arg-max of p / Exponential(1) noise in E, first maximum
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int j = tid; j < V; j += TT) {
        const float v = round_to<E>(prow[j] / E::to_f32(noise[(long)z * V + j]));
        if (v > best) { best = v; besti = j; }
    }
    const float gbest = block_reduce<E>(best, true, red);
    if (tid == 0) sh_besti = 0x7fffffff;
    __syncthreads();
    if (best == gbest) atomicMin(&sh_besti, besti);
    __syncthreads();
    if (tid == 0) {
        if (n_path < 15) path[n_path] = sh_besti == 0x7fffffff ? 0 : sh_besti;
        ++n_path;
        acc_num[z] = n_path;
        for (int i = 0; i < max_acc; ++i) acc_ids[(long)z * max_acc + i] = i < n_path ? path[i] : 0;
        words_used[z] = sh_bad ? -1 : sh_w;
    }
}

}  // namespace

extern "C" {

int ls_tree_grow(int64_t* tree_mask, int64_t* all_spec, float* logp_sum, const float* topk_vals, const int64_t* topk_idx,
                 int b, int F, int k, int64_t vocab, int lo, int mid, int32_t* base, int base_add, int64_t* positions,
                 uint32_t* bits, int words, void* stream) {
    if (!tree_mask || !all_spec || !logp_sum || !topk_vals || !topk_idx) LS_FAIL(LS_ERR_INVALID_ARG, "tree_grow: null pointer");
    if (b < 1 || F < 2 || F > MAXF || k < 1 || vocab < 1 || lo < 0 || mid <= lo || mid + k > F)
        LS_FAIL(LS_ERR_INVALID_ARG, "tree_grow: F=%d k=%d lo=%d mid=%d", F, k, lo, mid);
    if ((positions == nullptr) != (bits == nullptr)) LS_FAIL(LS_ERR_INVALID_ARG, "tree_grow: positions and bits go together");
    if (bits && words * 32 < mid + k) LS_FAIL(LS_ERR_INVALID_ARG, "tree_grow: %d words for %d columns", words, mid + k);
    hipLaunchKernelGGL(tree_grow_kernel, dim3(b), dim3(TT), 0, static_cast<hipStream_t>(stream), tree_mask, all_spec, logp_sum,
                       topk_vals, topk_idx, F, k, (long)vocab, lo, mid, base, base_add, positions, bits, words);
    LS_CHECK_LAUNCH("tree_grow_kernel");
    return LS_OK;
}

int ls_tree_verify_inputs(const int64_t* acc_ids, int64_t acc_stride, int a, const int64_t* all_spec, const int64_t* tree_mask,
                          int b, int F, int R, const int32_t* cache_lens, int64_t* veri_spec, int64_t* positions, uint32_t* bits,
                          int words, int32_t* bump, int bump_add, void* stream) {
    if (!acc_ids || !all_spec || !tree_mask || !veri_spec || !positions || !bits)
        LS_FAIL(LS_ERR_INVALID_ARG, "tree_verify_inputs: null pointer");
    if (b < 1 || F < 1 || F > MAXF || a < 1 || R < a + F - 1 || words * 32 < R)
        LS_FAIL(LS_ERR_INVALID_ARG, "tree_verify_inputs: a=%d F=%d R=%d words=%d", a, F, R, words);
    hipLaunchKernelGGL(tree_verify_inputs_kernel, dim3(b), dim3(TT), 0, static_cast<hipStream_t>(stream), acc_ids,
                       (long)acc_stride, a, all_spec, tree_mask, F, R, cache_lens, veri_spec, positions, bits, words, bump,
                       bump_add);
    LS_CHECK_LAUNCH("tree_verify_inputs_kernel");
    return LS_OK;
}

int ls_tree_collapse(const int64_t* all_spec, const int64_t* all_llm_pred, const int64_t* tree_mask,
                     const int32_t* cache_lens, int cache_len_add, int b, int F, int non_leaf_len, int max_acc,
                     int64_t* acc_ids, int64_t* acc_num, int32_t* double_input, int64_t* index_mapping, void* k_cache,
                     void* v_cache, int64_t kc_stride_b, int64_t kc_stride_s, int row_elems, int dtype, void* stream) {
    if (!all_spec || !all_llm_pred || !tree_mask || !acc_ids || !acc_num || !double_input || !index_mapping)
        LS_FAIL(LS_ERR_INVALID_ARG, "tree_collapse: null pointer");
    if (b < 1 || F < 1 || F > MAXF || max_acc < 1 || max_acc > F) LS_FAIL(LS_ERR_INVALID_ARG, "tree_collapse: F=%d max_acc=%d", F, max_acc);
    if (k_cache) {
        if (!v_cache || !cache_lens || row_elems < 8 || (row_elems & 7) || (kc_stride_s & 7))
            LS_FAIL(LS_ERR_INVALID_ARG, "tree_collapse: KV move args");
        if (dtype != LS_F16 && dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", dtype);
    }
    hipLaunchKernelGGL(tree_collapse_kernel, dim3(b), dim3(TT), 0, static_cast<hipStream_t>(stream), all_spec, all_llm_pred,
                       tree_mask, cache_lens, cache_len_add, F, non_leaf_len, max_acc, acc_ids, acc_num, double_input,
                       index_mapping, (char*)k_cache, (char*)v_cache, (long)kc_stride_b * 2, (long)kc_stride_s * 2,
                       row_elems * 2);
    LS_CHECK_LAUNCH("tree_collapse_kernel");
    return LS_OK;
}

int ls_tree_commit(const int64_t* acc_ids, const int64_t* acc_num, int b, int max_acc, int64_t* output_ids,
                   int64_t out_stride, int out_cap, int emitted, int32_t* emitted_dev, int has_eos, int64_t eos, int64_t* state,
                   int64_t* tree_mask, int64_t* all_spec, float* logp_sum, int F, int32_t* target_lens, int target_add,
                   int32_t* draft_kv_lens, void* stream) {
    if (!acc_ids || !acc_num || !output_ids || !state || !tree_mask || !all_spec)
        LS_FAIL(LS_ERR_INVALID_ARG, "tree_commit: null pointer");
    if (b < 1 || F < 1 || F > MAXF || max_acc < 1 || max_acc > TT || out_cap < 1 || emitted < 0)
        LS_FAIL(LS_ERR_INVALID_ARG, "tree_commit: F=%d max_acc=%d out_cap=%d emitted=%d", F, max_acc, out_cap, emitted);
    hipLaunchKernelGGL(tree_commit_kernel, dim3(b), dim3(TT), 0, static_cast<hipStream_t>(stream), acc_ids, acc_num, max_acc,
                       output_ids, (long)out_stride, out_cap, emitted, emitted_dev, has_eos, eos, state, tree_mask, all_spec,
                       logp_sum, F, target_lens, target_add, draft_kv_lens);
    LS_CHECK_LAUNCH("tree_commit_kernel");
    return LS_OK;
}

int ls_chain_commit(int64_t* llm_verify_output, int64_t* spec_buffer, int b, int gamma, int64_t* output_ids,
                    int64_t out_stride, int out_cap, int32_t* cache_lens, int32_t* draft_cache_lens, const int32_t* input_len,
                    int64_t* next_spec_start_token, int has_eos, int64_t eos, int64_t* state, const int64_t* accept_mask,
                    void* stream) {
    if (!llm_verify_output || !spec_buffer || !output_ids || !cache_lens || !draft_cache_lens || !input_len ||
        !next_spec_start_token || !state)
        LS_FAIL(LS_ERR_INVALID_ARG, "chain_commit: null pointer");
    if (b < 1 || gamma < 1 || gamma > 62 || out_cap < 1) LS_FAIL(LS_ERR_INVALID_ARG, "chain_commit: gamma=%d out_cap=%d", gamma, out_cap);
    hipLaunchKernelGGL(chain_commit_kernel, dim3(b), dim3(64), 0, static_cast<hipStream_t>(stream), llm_verify_output,
                       spec_buffer, gamma, output_ids, (long)out_stride, out_cap, cache_lens, draft_cache_lens, input_len,
                       next_spec_start_token, has_eos, eos, state, accept_mask);
    LS_CHECK_LAUNCH("chain_commit_kernel");
    return LS_OK;
}

int ls_embed_rows(const void* table, int64_t vocab, int hidden, int dtype, const int64_t* ids, int n, void* out, void* stream) {
    if (!table || !ids || !out || vocab < 1 || hidden < 8 || (hidden & 7)) LS_FAIL(LS_ERR_INVALID_ARG, "embed_rows args");
    if (dtype != LS_F16 && dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", dtype);
    if (n < 1) return LS_OK;
    hipLaunchKernelGGL(embed_rows_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), (const char*)table, ids,
                       (long)vocab, hidden * 2, (char*)out);
    LS_CHECK_LAUNCH("embed_rows_kernel");
    return LS_OK;
}


int ls_tree_verify_stochastic(const int64_t* all_spec, const int64_t* tree_mask, const void* llm_logits, int64_t logits_stride_b,
                              int64_t logits_stride_r, const float* spec_logp, int64_t logp_stride_b, int64_t logp_stride_r,
                              int b, int F, int Fs, int V, int dtype, float temperature, const uint32_t* mt_words, int n_words,
                              const void* exp_noise, int64_t* acc_ids, int max_acc, int64_t* acc_num, int32_t* words_used,
                              float* workspace, void* stream) {
    if (!all_spec || !tree_mask || !llm_logits || !spec_logp || !mt_words || !exp_noise || !acc_ids || !acc_num || !words_used ||
        !workspace)
        LS_FAIL(LS_ERR_INVALID_ARG, "verify_stochastic: null argument");
    if (b < 1 || F < 1 || F > MAXF || Fs < 1 || V < 1 || max_acc < 2 || max_acc > 15 || n_words < 3 || !(temperature > 0.f))
        LS_FAIL(LS_ERR_INVALID_ARG, "verify_stochastic: b=%d F=%d Fs=%d V=%d max_acc=%d n_words=%d T=%g", b, F, Fs, V, max_acc,
                n_words, (double)temperature);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LS_F16)
        hipLaunchKernelGGL(tree_verify_stochastic_kernel<ElemF16>, dim3(b), dim3(TT), 0, s, all_spec, tree_mask,
                           static_cast<const _Float16*>(llm_logits), (long)logits_stride_b, (long)logits_stride_r, spec_logp,
                           (long)logp_stride_b, (long)logp_stride_r, F, Fs, V, temperature, mt_words, n_words,
                           static_cast<const _Float16*>(exp_noise), acc_ids, max_acc, acc_num, words_used, workspace);
    else if (dtype == LS_BF16)
        hipLaunchKernelGGL(tree_verify_stochastic_kernel<ElemBF16>, dim3(b), dim3(TT), 0, s, all_spec, tree_mask,
                           static_cast<const __bf16*>(llm_logits), (long)logits_stride_b, (long)logits_stride_r, spec_logp,
                           (long)logp_stride_b, (long)logp_stride_r, F, Fs, V, temperature, mt_words, n_words,
                           static_cast<const __bf16*>(exp_noise), acc_ids, max_acc, acc_num, words_used, workspace);
    else
        LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", dtype);
    LS_CHECK_LAUNCH("tree_verify_stochastic_kernel");
    return LS_OK;
}

}  // extern "C"
