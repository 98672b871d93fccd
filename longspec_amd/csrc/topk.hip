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
// Vocabulary-parallel form (round 3, the lm_head sharded over the ranks of a node): the records are laid out CHUNK-major,
// [chunk slot][row][2 + 2k], so that the slots a rank owns are one contiguous block.  ls_topk_stage1 fills a rank's slots
// from its slice of the logits (columns reported as GLOBAL columns; slots beyond its slice get an empty record: max -inf, sum 0,
// no candidate), the blocks are all-gathered, ls_topk_stage2 merges all slots in slot order.  Slot order IS global chunk order
// and an empty record adds exactly +0.0f to the row sum, so values and indices are bit-identical to the one-GPU call for any
// number of ranks (stage 1 of a chunk never depended on the other chunks).
//
// Ties (equal fp32 values; fp16 logits collide often) go to the smaller flat index row * V + column, which
// is also what `argmax` returns in PyTorch.  Latency-bound (256 KB .. 18 MB of L2-resident logits):
// reported in microseconds, not against a roofline.
#include "ls_common.h"

namespace {

constexpr int TK_THREADS = 256;
constexpr int TK_CHUNK = 8192;          // logits per stage-1 workgroup: 32 per thread
constexpr int TK_MAXK = 64;          // also bounded by TK_THREADS (sentinel padding) and the merge kernel's registers

__device__ __forceinline__ bool better(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

// record of (chunk slot, row): float max, float sum, then k x (float value, int column), in no particular order; chunk-major
__device__ __forceinline__ float* ws_rec(float* ws, int row, int slot, int rows, int k) {
    return ws + ((long)slot * rows + row) * (2 + 2 * k);
}

// 16-bit float pattern -> unsigned key that orders like the value (no NaNs on this path)
__device__ __forceinline__ unsigned order_key(unsigned bits16) { return (bits16 & 0x8000u) ? (~bits16 & 0xffffu) : (bits16 | 0x8000u); }

// wave-wide max / min of a 32-bit unsigned through the DPP network (6 VALU pairs, no LDS crossbar): quad swaps,
// row rotations, then row broadcasts; every lane of the result's last row holds ... lane 63 holds the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);   // disabled lanes keep v
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = max(v, dpp_mov<0xB1, 0xf>(v));     // quad_perm [1,0,3,2]
    v = max(v, dpp_mov<0x4E, 0xf>(v));     // quad_perm [2,3,0,1]
    v = max(v, dpp_mov<0x124, 0xf>(v));    // row_ror:4
    v = max(v, dpp_mov<0x128, 0xf>(v));    // row_ror:8   -> every lane holds its row's max
    v = max(v, dpp_mov<0x142, 0xa>(v));    // row_bcast:15 into rows 1 and 3
    v = max(v, dpp_mov<0x143, 0xc>(v));    // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_umin(unsigned v) { return ~wave_umax(~v); }

// the same for a 64-bit key (value key in the high word, inverted flat index in the low word: one max = best
// value, ties to the smaller index)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max64(unsigned long long v) {
    const unsigned lo = dpp_mov<CTRL, ROW_MASK>((unsigned)v), hi = dpp_mov<CTRL, ROW_MASK>((unsigned)(v >> 32));
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_umax64(unsigned long long v) {
    v = dpp_max64<0xB1, 0xf>(v);
    v = dpp_max64<0x4E, 0xf>(v);
    v = dpp_max64<0x124, 0xf>(v);
    v = dpp_max64<0x128, 0xf>(v);
    v = dpp_max64<0x142, 0xa>(v);
    v = dpp_max64<0x143, 0xc>(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// fp32 -> unsigned key that orders like the value
__device__ __forceinline__ unsigned order_key32(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key32_value(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// Executed by wave 0 on a 256-bin histogram in LDS: the bin holding the `want`-th item counted from the top
// (from_top) or from the bottom, and how many items lie strictly beyond it on that side.
__device__ __forceinline__ void find_bin(const int* hist, int want, bool from_top, int lane, int& bin, int& beyond) {
    int c[4], tot = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = hist[from_top ? 255 - (lane * 4 + j) : lane * 4 + j];
        tot += c[j];
    }
    int incl = tot;                                   // inclusive prefix over lanes 0..lane (in scan direction)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    const int excl = incl - tot;
    const bool here = excl < want && incl >= want;    // exactly one lane
    int b = 0, bey = 0;
    if (here) {
        int run = excl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (run < want && run + c[j] >= want) { b = lane * 4 + j; bey = run; }
            run += c[j];
        }
    }
    const unsigned long long mask = __ballot(here);
    const int src = mask ? __ffsll((long long)mask) - 1 : 0;
    b = __shfl(b, src);
    bey = __shfl(bey, src);
    bin = from_top ? 255 - b : b;
    beyond = bey;
}

// Stage 1: one workgroup per (chunk of 8192 logits, row).  Exact selection of the chunk's k largest logits
// (ties to the smaller column) by two 256-bin histogram levels on the ordered 16-bit pattern -- and, only when
// the k-th value is shared by more elements than fit, two more levels on the column -- instead of k rounds of
// cross-lane arg-max (measured 2.5 us per round).
template <typename E>
__global__ __launch_bounds__(TK_THREADS) void topk_chunk_kernel(const typename E::T* __restrict__ logits, long ldl, int V, int k,
                                                                int col_base, float* __restrict__ ws) {
    // V = columns of THIS slice (local), col_base = global column of its column 0; grid.x = slots to fill, grid.y = rows
    __shared__ float s_red[8];
    __shared__ int s_hist[256];
    __shared__ int s_sel[4];          // [0] bin / threshold, [1] count beyond, [2] output cursor, [3] scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, row = blockIdx.y;
    const typename E::T* src = logits + (long)row * ldl;
    const int base = chunk * TK_CHUNK;
    const int n_valid = min(TK_CHUNK, V - base);
    if (n_valid <= 0) {               // a slot beyond this rank's slice of the vocabulary: the empty record
        float* rec = ws_rec(ws, row, chunk, gridDim.y, k);
        if (tid == 0) { rec[0] = -INFINITY; rec[1] = 0.f; }
        if (tid < k) { rec[2 + 2 * tid] = -INFINITY; reinterpret_cast<int*>(rec)[3 + 2 * tid] = 0x7fffffff; }
        return;
    }
    const int kk = min(k, n_valid);

    uint32_t raw[16];                 // 32 logits as 16-bit patterns; element e = j*8 + i sits at column col0[j] + i
    int col0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = base + (j * TK_THREADS + tid) * 8;
        col0[j] = c;
        uint4 v = make_uint4(0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u);      // -inf (fp16); masked by column anyway
        if (c < V) v = *reinterpret_cast<const uint4*>(src + c);                       // V % 8 == 0: all-in or all-out
        raw[j * 4 + 0] = v.x; raw[j * 4 + 1] = v.y; raw[j * 4 + 2] = v.z; raw[j * 4 + 3] = v.w;
    }
    auto bits = [&](int e) -> unsigned { return (raw[e >> 1] >> ((e & 1) * 16)) & 0xffffu; };
    auto val = [&](int e) -> float {
        const unsigned short h = (unsigned short)bits(e);
        return E::to_f32(__builtin_bit_cast(typename E::T, h));
    };
    auto valid = [&](int e) -> bool { return col0[e >> 3] < V; };
    s_hist[tid] = 0;
    if (tid == 0) s_sel[2] = 0;

    // ---- chunk max and sum of exp
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < 32; ++e)
        if (valid(e)) m = fmaxf(m, val(e));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e)
        if (valid(e)) s += expf(val(e) - m);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) s_red[4 + wave] = s;

    // ---- level 1: high byte of the key
#pragma unroll
    for (int e = 0; e < 32; ++e)
        if (valid(e)) atomicAdd(&s_hist[order_key(bits(e)) >> 8], 1);
    __syncthreads();
    if (wave == 0) {
        int b, bey;
        find_bin(s_hist, kk, true, lane, b, bey);
        if (lane == 0) { s_sel[0] = b; s_sel[1] = bey; }
    }
    __syncthreads();
    const int hi = s_sel[0], above_hi = s_sel[1];
    __syncthreads();
    s_hist[tid] = 0;
    __syncthreads();
    // ---- level 2: low byte among the elements of that bin
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const unsigned key = order_key(bits(e));
        if (valid(e) && (int)(key >> 8) == hi) atomicAdd(&s_hist[key & 255u], 1);
    }
    __syncthreads();
    if (wave == 0) {
        int b, bey;
        find_bin(s_hist, kk - above_hi, true, lane, b, bey);
        if (lane == 0) { s_sel[0] = (hi << 8) | b; s_sel[1] = above_hi + bey; s_sel[3] = s_hist[b]; }
    }
    __syncthreads();
    const unsigned T = (unsigned)s_sel[0];          // key of the kk-th largest logit
    const int n_gt = s_sel[1];                      // logits strictly larger
    const int n_eq = s_sel[3];                      // logits equal to it
    const int need = kk - n_gt;                     // how many of the equal ones belong to the top-k (>= 1)
    int col_limit = 0x7fffffff;                     // equal logits are taken up to this column
    if (need < n_eq) {                              // a tie at the boundary: the `need` smallest columns win
        __syncthreads();
        s_hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 32; ++e)
            if (valid(e) && order_key(bits(e)) == T) atomicAdd(&s_hist[(col0[e >> 3] + (e & 7) - base) >> 5], 1);
        __syncthreads();
        if (wave == 0) {
            int b, bey;
            find_bin(s_hist, need, false, lane, b, bey);
            if (lane == 0) { s_sel[0] = b; s_sel[1] = bey; }
        }
        __syncthreads();
        const int cb = s_sel[0], below = s_sel[1];
        __syncthreads();
        s_hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int rc = col0[e >> 3] + (e & 7) - base;
            if (valid(e) && order_key(bits(e)) == T && (rc >> 5) == cb) atomicAdd(&s_hist[rc & 31], 1);
        }
        __syncthreads();
        if (wave == 0) {
            int b, bey;
            find_bin(s_hist, need - below, false, lane, b, bey);
            if (lane == 0) s_sel[0] = base + cb * 32 + b;
        }
        __syncthreads();
        col_limit = s_sel[0];
    }
    // ---- emit exactly kk candidates (any order), pad with sentinels up to k
    float* rec = ws_rec(ws, row, chunk, gridDim.y, k);
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const unsigned key = order_key(bits(e));
        const int c = col0[e >> 3] + (e & 7);
        if (valid(e) && (key > T || (key == T && c <= col_limit))) {
            const int slot = atomicAdd(&s_sel[2], 1);
            rec[2 + 2 * slot] = val(e);
            reinterpret_cast<int*>(rec)[3 + 2 * slot] = col_base + c;
        }
    }
    if (tid < k - kk) {
        rec[2 + 2 * (kk + tid)] = -INFINITY;
        reinterpret_cast<int*>(rec)[3 + 2 * (kk + tid)] = 0x7fffffff;
    }
    if (tid == 0) {
        rec[0] = m;
        rec[1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    }
}

// mode 0: joint top-k over all rows of log_softmax(row) + history[row]; out_idx = row * V + column (sorted)
// mode 1: per-row argmax of the logits; out_idx[row] = column
// MAXOWN = candidates a thread holds in registers: ceil(R * nchunks * k / 256) rounded up to 4, 8, 16 or 20.  Every extraction round
// scans all MAXOWN slots of every thread, empty or not -- a 4-row level (1024 candidates) took as long as a 16-row one (round 5:
// the draft passes' merge 21 us for both; profiles/r5_round_timeline_128k.json).
template <int MAXOWN>
__global__ __launch_bounds__(TK_THREADS) void topk_merge_kernel(const float* __restrict__ ws, int R, int V, int k, int nchunks,
                                                                const float* __restrict__ history, int mode,
                                                                float* __restrict__ out_vals, int64_t* __restrict__ out_idx) {
    __shared__ float s_m[128], s_lse[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rec_f = 2 + 2 * k;
    if (mode == 1) {
        for (int row = blockIdx.x * TK_THREADS + tid; row < R; row += gridDim.x * TK_THREADS) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int c = 0; c < nchunks; ++c) {
                const float* rec = ws + ((long)c * R + row) * rec_f;
                const float v = rec[2];
                const int i = reinterpret_cast<const int*>(rec)[3];
                if (better(v, i, bv, bi)) { bv = v; bi = i; }
            }
            out_idx[row] = bi;
            if (out_vals) out_vals[row] = bv;
        }
        return;
    }
    // ---- per row: max and log-sum-exp in fixed chunk order.  All global loads of this kernel are issued in
    // independent batches (a dependent chain of L2 round trips per candidate cost 35 us in the first version).
    __shared__ float s_cm[128 * 4], s_cs[128 * 4], s_h[128];     // R * nchunks <= 512 (host-checked)
    for (int i = tid; i < R * nchunks; i += TK_THREADS) {         // i = row * nchunks + slot; records are slot-major
        const float2 ms = *reinterpret_cast<const float2*>(ws + ((long)(i % nchunks) * R + i / nchunks) * rec_f);
        s_cm[i] = ms.x;
        s_cs[i] = ms.y;
    }
    for (int row = tid; row < R; row += TK_THREADS) s_h[row] = history ? history[row] : 0.f;
    __syncthreads();
    for (int row = tid; row < R; row += TK_THREADS) {
        float m = -INFINITY;
        for (int c = 0; c < nchunks; ++c) m = fmaxf(m, s_cm[row * nchunks + c]);
        float s = 0.f;
        for (int c = 0; c < nchunks; ++c) s += s_cs[row * nchunks + c] * expf(s_cm[row * nchunks + c] - m);
        s_m[row] = m;
        s_lse[row] = logf(s);
    }
    __syncthreads();
    // ---- candidates: R * nchunks * k, strided over the threads and held in registers as (ordered key of the
    // value computed in the reference's operation order, flat index).  Each wave extracts its k best with DPP
    // reductions (no barrier), wave 0 then merges the 4 sorted lists: the output is sorted descending, ties to
    // the smaller flat index.
    const int ncand = R * nchunks * k;
    // ncand <= 256 * MAXOWN (the host picks the instantiation)
    unsigned long long pk[MAXOWN];      // (ordered key of the value) << 32 | ~flat index; 0 = empty
    float2 ld[MAXOWN];
    int crow[MAXOWN];
#pragma unroll
    for (int j = 0; j < MAXOWN; ++j) {
        const int id = min(j * TK_THREADS + tid, ncand - 1);           // clamped: every load is unconditional
        const int rc = id / k, slot = id - rc * k;                     // rc = chunk slot * R + row (the records' own order)
        crow[j] = rc % R;
        ld[j] = *reinterpret_cast<const float2*>(ws + (long)rc * rec_f + 2 + 2 * slot);
    }
#pragma unroll
    for (int j = 0; j < MAXOWN; ++j) {
        const int col = __float_as_int(ld[j].y);
        const int row = crow[j];
        float lp = (ld[j].x - s_m[row]) - s_lse[row];                  // log_softmax (llama_glide.py:1046)
        if (history) lp = lp + s_h[row];                               // + history_logp_sum (:1061)
        const unsigned long long key = ((unsigned long long)order_key32(lp) << 32) | (0xffffffffu - (unsigned)(row * V + col));
        pk[j] = (j * TK_THREADS + tid < ncand && col != 0x7fffffff) ? key : 0ull;     // rows * V < 2^31 (host-checked)
    }
    __shared__ unsigned long long s_w[4 * TK_MAXK];
    for (int r = 0; r < k; ++r) {
        unsigned long long b = pk[0];
#pragma unroll
        for (int j = 1; j < MAXOWN; ++j) b = pk[j] > b ? pk[j] : b;
        const unsigned long long w = wave_umax64(b);
        if (lane == 0) s_w[wave * TK_MAXK + r] = w;
#pragma unroll
        for (int j = 0; j < MAXOWN; ++j) pk[j] = pk[j] == w ? 0ull : pk[j];          // keys are unique
    }
    __syncthreads();
    if (wave == 0) {
        unsigned long long mk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = q * 64 + lane;
            mk[q] = id < 4 * k ? s_w[(id / k) * TK_MAXK + id % k] : 0ull;
        }
        for (int r = 0; r < k; ++r) {
            unsigned long long b = mk[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) b = mk[q] > b ? mk[q] : b;
            const unsigned long long w = wave_umax64(b);
            if (lane == 0) {
                out_vals[r] = key32_value((unsigned)(w >> 32));
                out_idx[r] = (int64_t)(0xffffffffu - (unsigned)w);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) mk[q] = mk[q] == w ? 0ull : mk[q];
        }
    }
}

}  // namespace

extern "C" {

size_t ls_topk_workspace_bytes(int rows, int vocab, int k) {
    if (rows < 1 || vocab < 8 || k < 1 || k > TK_MAXK) return 0;
    const int nchunks = (vocab + TK_CHUNK - 1) / TK_CHUNK;
    return (size_t)rows * nchunks * (2 + 2 * k) * sizeof(float);
}

static int topk_check(const void* logits, int rows, int vocab, int64_t ld, int dtype, int k, const char* what) {
    if (!logits) LS_FAIL(LS_ERR_INVALID_ARG, "%s: null pointer", what);
    if (dtype != LS_F16 && dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "%s: dtype", what);
    if (rows < 1 || vocab < 8 || vocab % 8 != 0 || ld < vocab || ld % 8 != 0)
        LS_FAIL(LS_ERR_UNSUPPORTED, "%s: rows=%d vocab=%d ld=%ld (vocab and ld must be multiples of 8)", what, rows, vocab, (long)ld);
    if (k < 1 || k > TK_MAXK) LS_FAIL(LS_ERR_UNSUPPORTED, "%s: k=%d (1..%d)", what, k, TK_MAXK);
    return LS_OK;
}

static int stage1(const void* logits, int rows, int vocab_local, int64_t ld, int dtype, int k, int col_base, int nslots, float* records,
                  hipStream_t s) {
    if (dtype == LS_F16)
        hipLaunchKernelGGL(topk_chunk_kernel<ElemF16>, dim3(nslots, rows), dim3(TK_THREADS), 0, s,
                           static_cast<const _Float16*>(logits), (long)ld, vocab_local, k, col_base, records);
    else
        hipLaunchKernelGGL(topk_chunk_kernel<ElemBF16>, dim3(nslots, rows), dim3(TK_THREADS), 0, s,
                           static_cast<const __bf16*>(logits), (long)ld, vocab_local, k, col_base, records);
    LS_CHECK_LAUNCH("topk_chunk_kernel");
    return LS_OK;
}

static int stage2(const float* records, int rows, int vocab, int k, int nslots, const float* history, int mode, float* out_vals,
                  int64_t* out_idx, hipStream_t s, const char* what) {
    if (!records || !out_idx) LS_FAIL(LS_ERR_INVALID_ARG, "%s: null pointer", what);
    if (mode == 0) {
        if (rows > 128) LS_FAIL(LS_ERR_UNSUPPORTED, "%s: more than 128 rows", what);
        if ((long)rows * nslots * k > 256L * 20) LS_FAIL(LS_ERR_UNSUPPORTED, "%s: rows*chunks*k too large", what);
        if ((long)rows * vocab < k) LS_FAIL(LS_ERR_INVALID_ARG, "%s: k exceeds the number of logits", what);
        if ((long)rows * vocab >= 0x7fffffffL) LS_FAIL(LS_ERR_UNSUPPORTED, "%s: rows * vocab >= 2^31", what);
        if ((long)rows * nslots > 512) LS_FAIL(LS_ERR_UNSUPPORTED, "%s: rows * chunks > 512", what);
        if (!out_vals) LS_FAIL(LS_ERR_INVALID_ARG, "%s: null out_vals", what);
    }
    const int grid2 = mode == 1 ? (rows + TK_THREADS - 1) / TK_THREADS : 1;
    const long ncand = mode == 1 ? 0 : (long)rows * nslots * k;
    if (ncand <= 256L * 4)
        hipLaunchKernelGGL(topk_merge_kernel<4>, dim3(grid2), dim3(TK_THREADS), 0, s, records, rows, vocab, k, nslots, history, mode,
                           out_vals, out_idx);
    else if (ncand <= 256L * 8)
        hipLaunchKernelGGL(topk_merge_kernel<8>, dim3(grid2), dim3(TK_THREADS), 0, s, records, rows, vocab, k, nslots, history, mode,
                           out_vals, out_idx);
    else if (ncand <= 256L * 16)
        hipLaunchKernelGGL(topk_merge_kernel<16>, dim3(grid2), dim3(TK_THREADS), 0, s, records, rows, vocab, k, nslots, history, mode,
                           out_vals, out_idx);
    else
        hipLaunchKernelGGL(topk_merge_kernel<20>, dim3(grid2), dim3(TK_THREADS), 0, s, records, rows, vocab, k, nslots, history, mode,
                           out_vals, out_idx);
    LS_CHECK_LAUNCH("topk_merge_kernel");
    return LS_OK;
}

static int topk_impl(const void* logits, int rows, int vocab, int64_t ld, int dtype, const float* history, int k, int mode,
                     float* out_vals, int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream, const char* what) {
    int rc = topk_check(logits, rows, vocab, ld, dtype, k, what);
    if (rc) return rc;
    if (!workspace) LS_FAIL(LS_ERR_INVALID_ARG, "%s: null pointer", what);
    const int nchunks = (vocab + TK_CHUNK - 1) / TK_CHUNK;
    if (workspace_bytes < ls_topk_workspace_bytes(rows, vocab, k)) LS_FAIL(LS_ERR_WORKSPACE, "%s: workspace too small", what);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* ws = static_cast<float*>(workspace);
    rc = stage1(logits, rows, vocab, ld, dtype, k, 0, nchunks, ws, s);
    if (rc) return rc;
    return stage2(ws, rows, vocab, k, nchunks, history, mode, out_vals, out_idx, s, what);
}

int ls_topk_chunk(void) { return TK_CHUNK; }

int ls_topk_stage1(const void* logits_local, int rows, int vocab_local, int64_t ld, int dtype, int k, int col_base, int nslots,
                   float* records, void* stream) {
    if (vocab_local == 0) {           // a rank that owns no column (more ranks than chunks): nothing but empty records
        if (!records || rows < 1 || k < 1 || k > TK_MAXK || nslots < 1 || (dtype != LS_F16 && dtype != LS_BF16))
            LS_FAIL(LS_ERR_INVALID_ARG, "ls_topk_stage1: bad arguments for an empty slice");
        return stage1(records, rows, 0, 8, dtype, k, col_base, nslots, records, static_cast<hipStream_t>(stream));
    }
    int rc = topk_check(logits_local, rows, vocab_local, ld, dtype, k, "ls_topk_stage1");
    if (rc) return rc;
    if (!records || nslots < 1 || col_base < 0 || col_base % TK_CHUNK != 0 || (long)nslots * TK_CHUNK < vocab_local)
        LS_FAIL(LS_ERR_INVALID_ARG, "ls_topk_stage1: records null, col_base %d not a multiple of %d, or %d slots < the slice's chunks",
                col_base, TK_CHUNK, nslots);
    return stage1(logits_local, rows, vocab_local, ld, dtype, k, col_base, nslots, records, static_cast<hipStream_t>(stream));
}

int ls_topk_stage2(const float* records, int rows, int vocab, int k, int nslots, const float* history, int argmax, float* out_vals,
                   int64_t* out_idx, void* stream) {
    if (rows < 1 || vocab < 8 || k < 1 || k > TK_MAXK || nslots < 1) LS_FAIL(LS_ERR_INVALID_ARG, "ls_topk_stage2: bad dims");
    return stage2(records, rows, vocab, k, nslots, history, argmax ? 1 : 0, out_vals, out_idx, static_cast<hipStream_t>(stream),
                  "ls_topk_stage2");
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
