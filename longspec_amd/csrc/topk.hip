// topk.hip -- beam-tree growth and greedy verification on the lm_head logits (gfx950 / CDNA4).
//
// Replaces, per draft pass, `lm_head(h).float().log_softmax(-1)`, `+ history_logp_sum[..., None]`,
// `.view(bsz, -1).topk(k)` (longspec/test/llama_glide.py:1019-1020,1046-1064: three full passes over the
// [rows, V] fp32 matrix plus a radix-sort based top-k, ~120 us of library kernels at V = 128256) and, per
// target pass, `lm_head(h).argmax(-1)` (llama_glide.py:1091), by two small launches that read the fp16
// logits once:
//
//   stage 1  grid (chunks of 8192 logits, rows): chunk max, chunk sum of exp(x - max), and the chunk's k
//            largest logits (value, column) -- within one row log-soft-max is monotone in the logit, so the
//            row's top-k by log-probability is its top-k by logit;
//   stage 2  one workgroup: per row m = max of chunk maxima, lse = log(sum of rescaled chunk sums) in
//            fixed chunk order, then the candidates' values ((x - m) - lse) + history[row] -- the
//            reference's operation order in fp32 -- and the global top-k, sorted descending.
//
// Ties (equal fp32 values; fp16 logits collide often) go to the smaller flat index row * V + column, which
// is also what `argmax` returns in PyTorch.  Latency-bound (256 KB .. 18 MB of L2-resident logits):
// reported in microseconds, not against a roofline.
#include "ls_common.h"

namespace {

constexpr int TK_THREADS = 256;
constexpr int TK_CHUNK = 8192;          // logits per stage-1 workgroup: 32 per thread
constexpr int TK_MAXK = 64;

struct Cand {
    float v;
    int i;
};
__device__ __forceinline__ bool better(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(c.v, off);
        const int oi = __shfl_xor(c.i, off);
        if (better(ov, oi, c.v, c.i)) { c.v = ov; c.i = oi; }
    }
    return c;
}

// ws layout per (row, chunk): float max, float sum, then k x (float value, int column)
__device__ __forceinline__ float* ws_rec(float* ws, int row, int chunk, int nchunks, int k) {
    return ws + ((long)row * nchunks + chunk) * (2 + 2 * k);
}

template <typename E>
__global__ __launch_bounds__(TK_THREADS) void topk_chunk_kernel(const typename E::T* __restrict__ logits, long ldl, int V, int k,
                                                                int nchunks, float* __restrict__ ws) {
    using V8 = typename E::V8;
    __shared__ float s_red[8];
    __shared__ Cand s_cand[4 * TK_MAXK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, row = blockIdx.y;
    const typename E::T* src = logits + (long)row * ldl;
    const int base = chunk * TK_CHUNK;

    float x[32];
    int col0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = base + (j * TK_THREADS + tid) * 8;
        col0[j] = c;
        if (c < V) {                                      // V % 8 == 0: a 16-byte vector is all-in or all-out
            const V8 v = *reinterpret_cast<const V8*>(src + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[j * 8 + e] = E::to_f32(v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[j * 8 + e] = -INFINITY;
        }
    }
    // ---- chunk max and sum of exp
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < 32; ++e) m = fmaxf(m, x[e]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) s += expf(x[e] - m);     // exp(-inf) = 0 for the padding
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) s_red[4 + wave] = s;

    // ---- k best of each wave (shuffles only), then of the 4 waves
    Cand mine;
    auto rescan = [&]() {
        mine.v = -INFINITY;
        mine.i = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int c = col0[e >> 3] + (e & 7);
            if (x[e] != -INFINITY && better(x[e], c, mine.v, mine.i)) { mine.v = x[e]; mine.i = c; }   // -inf = padding / retired
        }
    };
    rescan();
    for (int r = 0; r < k; ++r) {
        const Cand b = wave_best(mine);
        if (lane == 0) s_cand[wave * TK_MAXK + r] = b;
        if (b.i == mine.i && b.v == mine.v && b.i != 0x7fffffff) {   // this lane owned the winner: retire it
#pragma unroll
            for (int e = 0; e < 32; ++e)
                if (col0[e >> 3] + (e & 7) == b.i) x[e] = -INFINITY;
            rescan();
        }
    }
    __syncthreads();
    if (wave == 0) {
        float* rec = ws_rec(ws, row, chunk, nchunks, k);
        if (lane == 0) {
            rec[0] = m;
            rec[1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
        }
        // 4k candidates, up to 4 per lane
        Cand c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = q * 64 + lane;
            const int w = id / k, r = id % k;
            if (id < 4 * k) c[q] = s_cand[w * TK_MAXK + r];
            else { c[q].v = -INFINITY; c[q].i = 0x7fffffff; }
        }
        for (int r = 0; r < k; ++r) {
            Cand l = c[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (better(c[q].v, c[q].i, l.v, l.i)) l = c[q];
            const Cand b = wave_best(l);
            if (lane == 0) {
                rec[2 + 2 * r] = b.v;
                reinterpret_cast<int*>(rec)[3 + 2 * r] = b.i;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c[q].i == b.i && c[q].v == b.v) { c[q].v = -INFINITY; c[q].i = 0x7fffffff; }
        }
    }
}

// mode 0: joint top-k over all rows of log_softmax(row) + history[row]; out_idx = row * V + column (sorted)
// mode 1: per-row argmax of the logits; out_idx[row] = column
__global__ __launch_bounds__(TK_THREADS) void topk_merge_kernel(const float* __restrict__ ws, int R, int V, int k, int nchunks,
                                                                const float* __restrict__ history, int mode,
                                                                float* __restrict__ out_vals, int64_t* __restrict__ out_idx) {
    __shared__ float s_m[128], s_lse[128];
    __shared__ float s_v[4];
    __shared__ long s_f[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rec_f = 2 + 2 * k;
    if (mode == 1) {
        for (int row = blockIdx.x * TK_THREADS + tid; row < R; row += gridDim.x * TK_THREADS) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int c = 0; c < nchunks; ++c) {
                const float* rec = ws + ((long)row * nchunks + c) * rec_f;
                const float v = rec[2];
                const int i = reinterpret_cast<const int*>(rec)[3];
                if (better(v, i, bv, bi)) { bv = v; bi = i; }
            }
            out_idx[row] = bi;
            if (out_vals) out_vals[row] = bv;
        }
        return;
    }
    // ---- per row: max and log-sum-exp in fixed chunk order
    for (int row = tid; row < R; row += TK_THREADS) {
        float m = -INFINITY;
        for (int c = 0; c < nchunks; ++c) m = fmaxf(m, ws[((long)row * nchunks + c) * rec_f]);
        float s = 0.f;
        for (int c = 0; c < nchunks; ++c) {
            const float* rec = ws + ((long)row * nchunks + c) * rec_f;
            s += rec[1] * expf(rec[0] - m);
        }
        s_m[row] = m;
        s_lse[row] = logf(s);
    }
    __syncthreads();
    // ---- candidates: R * nchunks * k, strided over the threads; value in the reference's operation order
    const int ncand = R * nchunks * k;
    auto cand_at = [&](int id, float& v, long& flat) {
        const int row = id / (nchunks * k), rem = id % (nchunks * k);
        const float* rec = ws + ((long)row * nchunks + rem / k) * rec_f;
        const float x = rec[2 + 2 * (rem % k)];
        const int col = reinterpret_cast<const int*>(rec)[3 + 2 * (rem % k)];
        float lp = (x - s_m[row]) - s_lse[row];                       // log_softmax (llama_glide.py:1046)
        if (history) lp = lp + history[row];                          // + history_logp_sum (:1061)
        v = col == 0x7fffffff ? -INFINITY : lp;
        flat = col == 0x7fffffff ? 0x7fffffffffffffffL : (long)row * V + col;
    };
    // each thread scans its candidates every round, skipping retired ones (k, ncand/256 are small)
    constexpr int MAXOWN = 40;          // ncand <= 256 * 40
    unsigned long long retired = 0ull;  // bit j: this thread's j-th candidate is out
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        long bf = 0x7fffffffffffffffL;
        int bj = -1;
        for (int j = 0, id = tid; id < ncand && j < MAXOWN; ++j, id += TK_THREADS) {
            if (retired >> j & 1ull) continue;
            float v;
            long f;
            cand_at(id, v, f);
            if (v > bv || (v == bv && f < bf)) { bv = v; bf = f; bj = j; }
        }
        // workgroup arg-best over (bv, bf)
        float wv = bv;
        long wf = bf;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(wv, off);
            const long of = __shfl_xor(wf, off);
            if (ov > wv || (ov == wv && of < wf)) { wv = ov; wf = of; }
        }
        if (lane == 0) {
            s_v[wave] = wv;
            s_f[wave] = wf;
        }
        __syncthreads();
        float gv = s_v[0];
        long gf = s_f[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_v[w] > gv || (s_v[w] == gv && s_f[w] < gf)) { gv = s_v[w]; gf = s_f[w]; }
        if (tid == 0) {
            out_vals[r] = gv;
            out_idx[r] = gf;
        }
        if (bj >= 0 && bf == gf) retired |= 1ull << bj;
        __syncthreads();
    }
}

}  // namespace

extern "C" {

size_t ls_topk_workspace_bytes(int rows, int vocab, int k) {
    if (rows < 1 || vocab < 8 || k < 1 || k > TK_MAXK) return 0;
    const int nchunks = (vocab + TK_CHUNK - 1) / TK_CHUNK;
    return (size_t)rows * nchunks * (2 + 2 * k) * sizeof(float);
}

static int topk_impl(const void* logits, int rows, int vocab, int64_t ld, int dtype, const float* history, int k, int mode,
                     float* out_vals, int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream, const char* what) {
    if (!logits || !out_idx || !workspace) LS_FAIL(LS_ERR_INVALID_ARG, "%s: null pointer", what);
    if (dtype != LS_F16 && dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "%s: dtype", what);
    if (rows < 1 || vocab < 8 || vocab % 8 != 0 || ld < vocab || ld % 8 != 0)
        LS_FAIL(LS_ERR_UNSUPPORTED, "%s: rows=%d vocab=%d ld=%ld (vocab and ld must be multiples of 8)", what, rows, vocab, (long)ld);
    if (k < 1 || k > TK_MAXK) LS_FAIL(LS_ERR_UNSUPPORTED, "%s: k=%d (1..%d)", what, k, TK_MAXK);
    const int nchunks = (vocab + TK_CHUNK - 1) / TK_CHUNK;
    if (mode == 0) {
        if (rows > 128) LS_FAIL(LS_ERR_UNSUPPORTED, "%s: more than 128 rows", what);
        if ((long)rows * nchunks * k > 256L * 40) LS_FAIL(LS_ERR_UNSUPPORTED, "%s: rows*chunks*k too large", what);
        if ((long)rows * vocab < k) LS_FAIL(LS_ERR_INVALID_ARG, "%s: k exceeds the number of logits", what);
        if (!out_vals) LS_FAIL(LS_ERR_INVALID_ARG, "%s: null out_vals", what);
    }
    if (workspace_bytes < ls_topk_workspace_bytes(rows, vocab, k)) LS_FAIL(LS_ERR_WORKSPACE, "%s: workspace too small", what);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* ws = static_cast<float*>(workspace);
    if (dtype == LS_F16)
        hipLaunchKernelGGL(topk_chunk_kernel<ElemF16>, dim3(nchunks, rows), dim3(TK_THREADS), 0, s,
                           static_cast<const _Float16*>(logits), (long)ld, vocab, k, nchunks, ws);
    else
        hipLaunchKernelGGL(topk_chunk_kernel<ElemBF16>, dim3(nchunks, rows), dim3(TK_THREADS), 0, s,
                           static_cast<const __bf16*>(logits), (long)ld, vocab, k, nchunks, ws);
    LS_CHECK_LAUNCH("topk_chunk_kernel");
    const int grid2 = mode == 1 ? (rows + TK_THREADS - 1) / TK_THREADS : 1;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(grid2), dim3(TK_THREADS), 0, s, ws, rows, vocab, k, nchunks, history, mode, out_vals,
                       out_idx);
    LS_CHECK_LAUNCH("topk_merge_kernel");
    return LS_OK;
}

int ls_logprob_topk(const void* logits, int rows, int vocab, int64_t ld, int dtype, const float* history, int k, float* out_vals,
                    int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
    return topk_impl(logits, rows, vocab, ld, dtype, history, k, 0, out_vals, out_idx, workspace, workspace_bytes, stream,
                     "ls_logprob_topk");
}

int ls_argmax_rows(const void* logits, int rows, int vocab, int64_t ld, int dtype, int64_t* out_idx, void* workspace,
                   size_t workspace_bytes, void* stream) {
    return topk_impl(logits, rows, vocab, ld, dtype, nullptr, 1, 1, nullptr, out_idx, workspace, workspace_bytes, stream,
                     "ls_argmax_rows");
}

}  // extern "C"
