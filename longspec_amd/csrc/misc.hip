// Latency-bound operators of the decode round: RMSNorm, RoPE and tree positions (the beam-tree
// bookkeeping kernels live in tree.hip).  KB-scale traffic each: single-pass kernels, vectorised
// 16-byte accesses, no host synchronisation.
#include <stdarg.h>

#include "ls_common.h"

// ---- error plumbing ---------------------------------------------------------------
static thread_local char g_err[512] = "";
void ls_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

// ---- RMSNorm --------------------------------------------------------------------------
// LlamaRMSNorm.forward: y = w * dtype(x32 * rsqrt(mean(x32^2) + eps)); optional fused
// `residual + hidden` (rounded to dtype first, llama.py:492,498).
template <typename E>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const typename E::T* __restrict__ x,
                                                      const typename E::T* __restrict__ res,
                                                      const typename E::T* __restrict__ wgt, typename E::T* __restrict__ y,
                                                      typename E::T* __restrict__ sum_out, int hidden, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typename E::T* row = reinterpret_cast<typename E::T*>(smem);          // the (summed) row, dtype
    float* red = reinterpret_cast<float*>(smem + ((hidden * 2 + 15) & ~15));
    const long base = (long)blockIdx.x * hidden;
    const int tid = threadIdx.x;
    float ss = 0.f;
    for (int i = tid * 8; i < hidden; i += 256 * 8) {
        typename E::V8 v = *reinterpret_cast<const typename E::V8*>(x + base + i);
        if (res) {
            typename E::V8 r = *reinterpret_cast<const typename E::V8*>(res + base + i);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = E::from_f32(E::to_f32(r[e]) + E::to_f32(v[e]));
            if (sum_out) *reinterpret_cast<typename E::V8*>(sum_out + base + i) = v;
        }
        *reinterpret_cast<typename E::V8*>(row + i) = v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = E::to_f32(v[e]);
            ss += f * f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float rs = rsqrtf(tot / (float)hidden + eps);
    for (int i = tid * 8; i < hidden; i += 256 * 8) {
        typename E::V8 v = *reinterpret_cast<const typename E::V8*>(row + i);
        typename E::V8 w8 = *reinterpret_cast<const typename E::V8*>(wgt + i);
        typename E::V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float n = round_to<E>(E::to_f32(v[e]) * rs);
            o[e] = E::from_f32(E::to_f32(w8[e]) * n);
        }
        *reinterpret_cast<typename E::V8*>(y + base + i) = o;
    }
}

// Short-chain variant for the decode passes (<= 80 rows, a launch is pure latency): one 8-element chunk per thread
// (NCH of them for hidden > 8192), every global load -- x, residual AND weight -- issued before the first use,
// the row kept in registers, one barrier.
// CANON (hidden a multiple of 64): the sum of squares is taken in the canonical order of ls_common.h (quads, 16-column
// tiles, 64-column slabs, groups of eight slabs in column order) -- the order in which the projections of gemm.hip produce and consume it, so
// that a norm folded into the next projection and this kernel give the same bits.
template <typename E, int NCH, bool CANON>
__device__ __forceinline__ void rmsnorm_row(const typename E::T* __restrict__ xrow, const typename E::T* __restrict__ resrow,
                                            const typename E::T* __restrict__ wgt, typename E::T* __restrict__ yrow,
                                            typename E::T* __restrict__ sumrow, typename E::T* __restrict__ copyrow, int hidden,
                                            float eps) {
    // xrow / resrow / yrow / sumrow / copyrow: THIS row (workgroup) already; `copyrow`: x as loaded (pass_head: the gathered
    // embedding row, the pass's residual stream)
    __shared__ float red[16];
    const int tid = threadIdx.x, T = blockDim.x;
    typename E::V8 v[NCH], r[NCH], w8[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * T + tid) * 8;
        if (i < hidden) {
            v[c] = *reinterpret_cast<const typename E::V8*>(xrow + i);
            if (resrow) r[c] = *reinterpret_cast<const typename E::V8*>(resrow + i);
            w8[c] = *reinterpret_cast<const typename E::V8*>(wgt + i);
        }
    }
    __shared__ float slab_s[CANON ? 512 : 1];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * T + tid) * 8;
        float s8 = 0.f;
        if (i < hidden) {
            if (copyrow) *reinterpret_cast<typename E::V8*>(copyrow + i) = v[c];
            if (resrow) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = E::from_f32(E::to_f32(r[c][e]) + E::to_f32(v[c][e]));
                if (sumrow) *reinterpret_cast<typename E::V8*>(sumrow + i) = v[c];
            }
            if (CANON) {
                s8 = ssq_quad(E::to_f32(v[c][0]), E::to_f32(v[c][1]), E::to_f32(v[c][2]), E::to_f32(v[c][3])) +
                     ssq_quad(E::to_f32(v[c][4]), E::to_f32(v[c][5]), E::to_f32(v[c][6]), E::to_f32(v[c][7]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = E::to_f32(v[c][e]);
                    ss += f * f;
                }
            }
        }
        if (CANON) {             // 2 threads = a 16-column tile, 8 threads = a 64-column slab (whole slabs are valid or not)
            const float tile = s8 + __shfl_xor(s8, 1);
            const int b8 = (tid & 63) & ~7;
            const float slab = ssq_slab64(__shfl(tile, b8), __shfl(tile, b8 + 2), __shfl(tile, b8 + 4), __shfl(tile, b8 + 6));
            if ((tid & 7) == 0 && i < hidden) slab_s[i >> 6] = slab;
        }
    }
    float tot = 0.f;
    if (CANON) {
        __syncthreads();
        tot = ssq_row(slab_s, hidden >> 6);
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        for (int wv = 0; wv < (T >> 6); ++wv) tot += red[wv];
    }
    const float rs = rsqrtf(tot / (float)hidden + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = (c * T + tid) * 8;
        if (i < hidden) {
            typename E::V8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = round_to<E>(E::to_f32(v[c][e]) * rs);
                o[e] = E::from_f32(E::to_f32(w8[c][e]) * n);
            }
            *reinterpret_cast<typename E::V8*>(yrow + i) = o;
        }
    }
}

template <typename E, int NCH, bool CANON>
__global__ __launch_bounds__(1024) void rmsnorm_rows_kernel(const typename E::T* __restrict__ x,
                                                            const typename E::T* __restrict__ res,
                                                            const typename E::T* __restrict__ wgt, typename E::T* __restrict__ y,
                                                            typename E::T* __restrict__ sum_out, int hidden, float eps) {
    const long base = (long)blockIdx.x * hidden;
    rmsnorm_row<E, NCH, CANON>(x + base, res ? res + base : nullptr, wgt, y + base, sum_out ? sum_out + base : nullptr, nullptr, hidden, eps);
}

// cos/sin of one position, element j of 64: shared by rope_cos_sin_kernel and pass_head_kernel (the same bits)
template <typename E>
__device__ __forceinline__ void rope_row_entry(float posf, const float* __restrict__ inv_freq, float scaling, int j,
                                               typename E::T* __restrict__ cosrow, typename E::T* __restrict__ sinrow) {
    const float f = posf * inv_freq[j];
    const float c = (float)cos((double)f) * scaling;
    const float s = (float)sin((double)f) * scaling;
    const typename E::T ce = E::from_f32(c), se = E::from_f32(s);
    cosrow[j] = ce;
    cosrow[64 + j] = ce;
    sinrow[j] = se;
    sinrow[64 + j] = se;
}

// Head of a decode pass in ONE launch (round 6): embedding gather (tree.hip: embed_rows_kernel) + the RoPE table of the pass
// (rope_cos_sin_kernel) + the first layer's input RMSNorm (rmsnorm_rows_kernel) -- three 4-7 us latency-bound launches in front
// of every draft pass, the verify pass and a vanilla step (profiles/r5_round_timeline_128k.json: 16.5 us per pass).  One
// workgroup per token row; the same device functions as the three kernels, so the outputs are bit-identical to theirs.
//   position of row i: positions[i] if given, else pos_base[i / q_len] + i % q_len + pos_add (the `arange + cache_lens[:, None]`
//   of llama_glide.py:1005 / llama.py:571-577, without the elementwise launch).
template <typename E, int NCH, bool CANON>
__global__ __launch_bounds__(1024) void pass_head_kernel(const typename E::T* __restrict__ table, const int64_t* __restrict__ ids,
                                                         long vocab, const int64_t* __restrict__ positions,
                                                         const int32_t* __restrict__ pos_base, int q_len, int pos_add,
                                                         const float* __restrict__ inv_freq, float scaling,
                                                         const typename E::T* __restrict__ wgt, typename E::T* __restrict__ embeds,
                                                         typename E::T* __restrict__ y, typename E::T* __restrict__ cosv,
                                                         typename E::T* __restrict__ sinv, int hidden, float eps) {
    const int row = blockIdx.x;
    long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // memory safety, as embed_rows_kernel
    if (threadIdx.x < 64) {
        const long pos = positions ? positions[row] : (long)pos_base[row / q_len] + row % q_len + pos_add;
        rope_row_entry<E>((float)pos, inv_freq, scaling, threadIdx.x, cosv + (long)row * 128, sinv + (long)row * 128);
    }
    const long base = (long)row * hidden;
    rmsnorm_row<E, NCH, CANON>(table + id * (long)hidden, nullptr, wgt, y + base, nullptr, embeds + base, hidden, eps);
}

// ---- RoPE -------------------------------------------------------------------------------
// cos/sin[r][d] = dtype( {cos,sin}( float(pos[r]) * inv_freq[d % 64] ) * scaling )
// (fp32 product like the reference's K=1 matmul; the transcendental is evaluated in
// double and rounded once to fp32 so the table does not depend on a fast-math libm.)
template <typename E>
__global__ void rope_cos_sin_kernel(const int64_t* __restrict__ pos, const float* __restrict__ inv_freq, float scaling,
                                    typename E::T* __restrict__ cosv, typename E::T* __restrict__ sinv, int rows) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * 64) return;
    const int r = idx >> 6, j = idx & 63;
    rope_row_entry<E>((float)pos[r], inv_freq, scaling, j, cosv + (long)r * 128, sinv + (long)r * 128);
}

// x*cos + rotate_half(x)*sin, each product and the sum rounded to dtype.
// One thread = 8 contiguous elements d0..d0+7 of the low half AND the matching 8 of the high half.
template <typename E>
__global__ void rope_apply_kernel(typename E::T* __restrict__ q, typename E::T* __restrict__ k,
                                  const typename E::T* __restrict__ cosv, const typename E::T* __restrict__ sinv, int rows,
                                  int Hq, int Hk, long q_rs, long k_rs) {
    const int per_row = (Hq + Hk) * 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * per_row) return;
    const int r = (int)(idx / per_row);
    const int rem = (int)(idx % per_row);
    const int h = rem >> 3, d0 = (rem & 7) * 8;
    typename E::T* base = h < Hq ? q + (long)r * q_rs + (long)h * 128 : k + (long)r * k_rs + (long)(h - Hq) * 128;
    typename E::V8 lo = *reinterpret_cast<typename E::V8*>(base + d0);
    typename E::V8 hi = *reinterpret_cast<typename E::V8*>(base + 64 + d0);
    typename E::V8 cl = *reinterpret_cast<const typename E::V8*>(cosv + (long)r * 128 + d0);
    typename E::V8 ch = *reinterpret_cast<const typename E::V8*>(cosv + (long)r * 128 + 64 + d0);
    typename E::V8 sl = *reinterpret_cast<const typename E::V8*>(sinv + (long)r * 128 + d0);
    typename E::V8 sh = *reinterpret_cast<const typename E::V8*>(sinv + (long)r * 128 + 64 + d0);
    typename E::V8 ol, oh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xl = E::to_f32(lo[e]), xh = E::to_f32(hi[e]);
        // low half: x*cos + (-x_high)*sin ; high half: x*cos + x_low*sin
        ol[e] = E::from_f32(round_to<E>(xl * E::to_f32(cl[e])) + round_to<E>(-xh * E::to_f32(sl[e])));
        oh[e] = E::from_f32(round_to<E>(xh * E::to_f32(ch[e])) + round_to<E>(xl * E::to_f32(sh[e])));
    }
    *reinterpret_cast<typename E::V8*>(base + d0) = ol;
    *reinterpret_cast<typename E::V8*>(base + 64 + d0) = oh;
}

__global__ void tree_positions_kernel(const int64_t* __restrict__ mask, const int32_t* __restrict__ base, int M, int N,
                                      int64_t* __restrict__ pos, int total_rows) {
    const int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= total_rows) return;
    long s = 0;
    for (int j = lane; j < N; j += 64) s += mask[(long)row * N + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) pos[row] = s - 1 + (base ? (long)base[row / M] : 0);
}

}  // namespace

extern "C" {

int ls_version(void) { return 100; }
const char* ls_last_error(void) { return g_err; }

int ls_rmsnorm_fwd(const void* x, const void* residual, const void* weight, void* y, void* sum_out, int rows, int hidden,
                   float eps, int dtype, void* stream) {
    if (!x || !weight || !y || rows < 1 || hidden < 8 || (hidden & 7)) LS_FAIL(LS_ERR_INVALID_ARG, "rmsnorm args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype != LS_F16 && dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", dtype);
    const int chunks = hidden / 8;
    if (rows <= 128 && chunks <= 4096) {
        // decode-shaped call: chunk-per-thread kernel
        int T = ((chunks + 63) / 64) * 64;
        int nch = 1;
        if (T > 1024) {
            nch = chunks <= 2048 ? 2 : 4;
            T = (((chunks + nch - 1) / nch + 63) / 64) * 64;
        }
#define LS_NORM_ROWS(EL, TY, N)                                                                                           \
    if (hidden % 64 == 0)                                                                                                 \
        hipLaunchKernelGGL((rmsnorm_rows_kernel<EL, N, true>), dim3(rows), dim3(T), 0, s, (const TY*)x,                   \
                           (const TY*)residual, (const TY*)weight, (TY*)y, (TY*)sum_out, hidden, eps);                    \
    else                                                                                                                  \
        hipLaunchKernelGGL((rmsnorm_rows_kernel<EL, N, false>), dim3(rows), dim3(T), 0, s, (const TY*)x,                  \
                           (const TY*)residual, (const TY*)weight, (TY*)y, (TY*)sum_out, hidden, eps)
        if (dtype == LS_F16) {
            if (nch == 1) { LS_NORM_ROWS(ElemF16, _Float16, 1); }
            else if (nch == 2) { LS_NORM_ROWS(ElemF16, _Float16, 2); }
            else { LS_NORM_ROWS(ElemF16, _Float16, 4); }
        } else {
            if (nch == 1) { LS_NORM_ROWS(ElemBF16, __bf16, 1); }
            else if (nch == 2) { LS_NORM_ROWS(ElemBF16, __bf16, 2); }
            else { LS_NORM_ROWS(ElemBF16, __bf16, 4); }
        }
#undef LS_NORM_ROWS
        LS_CHECK_LAUNCH("rmsnorm_rows_kernel");
        return LS_OK;
    }
    const size_t lds = ((hidden * 2 + 15) & ~15) + 16;
    if (lds > 64 * 1024) LS_FAIL(LS_ERR_UNSUPPORTED, "hidden %d too large", hidden);
    if (dtype == LS_F16)
        hipLaunchKernelGGL(rmsnorm_kernel<ElemF16>, dim3(rows), dim3(256), lds, s, (const _Float16*)x,
                           (const _Float16*)residual, (const _Float16*)weight, (_Float16*)y, (_Float16*)sum_out, hidden, eps);
    else
        hipLaunchKernelGGL(rmsnorm_kernel<ElemBF16>, dim3(rows), dim3(256), lds, s, (const __bf16*)x,
                           (const __bf16*)residual, (const __bf16*)weight, (__bf16*)y, (__bf16*)sum_out, hidden, eps);
    LS_CHECK_LAUNCH("rmsnorm_kernel");
    return LS_OK;
}

int ls_pass_head(const void* table, int64_t vocab, int hidden, int dtype, const int64_t* ids, int rows, const int64_t* positions,
                 const int32_t* pos_base, int q_len, int pos_add, const float* inv_freq, float attention_scaling,
                 const void* norm_weight, float eps, void* embeds, void* normed, void* cosv, void* sinv, void* stream) {
    if (!table || !ids || !inv_freq || !norm_weight || !embeds || !normed || !cosv || !sinv || vocab < 1 || rows < 1 || rows > 128 ||
        hidden < 8 || (hidden & 7) || hidden / 8 > 4096 || (!positions && (!pos_base || q_len < 1)))
        LS_FAIL(LS_ERR_INVALID_ARG, "pass_head args (rows %d <= 128, hidden %d)", rows, hidden);
    if (dtype != LS_F16 && dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", dtype);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int chunks = hidden / 8;
    int T = ((chunks + 63) / 64) * 64;          // (>= 64: the first wave also writes the row's cos/sin)
    int nch = 1;
    if (T > 1024) {
        nch = chunks <= 2048 ? 2 : 4;
        T = (((chunks + nch - 1) / nch + 63) / 64) * 64;
    }
#define LS_PASS_HEAD(EL, TY, N)                                                                                           \
    if (hidden % 64 == 0)                                                                                                 \
        hipLaunchKernelGGL((pass_head_kernel<EL, N, true>), dim3(rows), dim3(T), 0, s, (const TY*)table, ids, (long)vocab, \
                           positions, pos_base, q_len, pos_add, inv_freq, attention_scaling, (const TY*)norm_weight,      \
                           (TY*)embeds, (TY*)normed, (TY*)cosv, (TY*)sinv, hidden, eps);                                  \
    else                                                                                                                  \
        hipLaunchKernelGGL((pass_head_kernel<EL, N, false>), dim3(rows), dim3(T), 0, s, (const TY*)table, ids, (long)vocab, \
                           positions, pos_base, q_len, pos_add, inv_freq, attention_scaling, (const TY*)norm_weight,      \
                           (TY*)embeds, (TY*)normed, (TY*)cosv, (TY*)sinv, hidden, eps)
    if (dtype == LS_F16) {
        if (nch == 1) { LS_PASS_HEAD(ElemF16, _Float16, 1); }
        else if (nch == 2) { LS_PASS_HEAD(ElemF16, _Float16, 2); }
        else { LS_PASS_HEAD(ElemF16, _Float16, 4); }
    } else {
        if (nch == 1) { LS_PASS_HEAD(ElemBF16, __bf16, 1); }
        else if (nch == 2) { LS_PASS_HEAD(ElemBF16, __bf16, 2); }
        else { LS_PASS_HEAD(ElemBF16, __bf16, 4); }
    }
#undef LS_PASS_HEAD
    LS_CHECK_LAUNCH("pass_head_kernel");
    return LS_OK;
}

int ls_rope_cos_sin(const int64_t* positions, const float* inv_freq, float attention_scaling, void* cosv, void* sinv,
                    int rows, int dtype, void* stream) {
    if (!positions || !inv_freq || !cosv || !sinv || rows < 1) LS_FAIL(LS_ERR_INVALID_ARG, "rope_cos_sin args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n = rows * 64;
    if (dtype == LS_F16)
        hipLaunchKernelGGL(rope_cos_sin_kernel<ElemF16>, dim3((n + 255) / 256), dim3(256), 0, s, positions, inv_freq,
                           attention_scaling, (_Float16*)cosv, (_Float16*)sinv, rows);
    else if (dtype == LS_BF16)
        hipLaunchKernelGGL(rope_cos_sin_kernel<ElemBF16>, dim3((n + 255) / 256), dim3(256), 0, s, positions, inv_freq,
                           attention_scaling, (__bf16*)cosv, (__bf16*)sinv, rows);
    else
        LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", dtype);
    LS_CHECK_LAUNCH("rope_cos_sin_kernel");
    return LS_OK;
}

int ls_rope_apply(void* q, void* k, const void* cosv, const void* sinv, int rows, int Hq, int Hk, int64_t q_row_stride,
                  int64_t k_row_stride, int dtype, void* stream) {
    if (!q || !cosv || !sinv || rows < 1 || Hq < 1 || Hk < 0 || (Hk > 0 && !k)) LS_FAIL(LS_ERR_INVALID_ARG, "rope_apply args");
    if ((q_row_stride & 7) || (k_row_stride & 7)) LS_FAIL(LS_ERR_INVALID_ARG, "row strides must be multiples of 8");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long n = (long)rows * (Hq + Hk) * 8;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == LS_F16)
        hipLaunchKernelGGL(rope_apply_kernel<ElemF16>, grid, dim3(256), 0, s, (_Float16*)q, (_Float16*)k,
                           (const _Float16*)cosv, (const _Float16*)sinv, rows, Hq, Hk, (long)q_row_stride, (long)k_row_stride);
    else if (dtype == LS_BF16)
        hipLaunchKernelGGL(rope_apply_kernel<ElemBF16>, grid, dim3(256), 0, s, (__bf16*)q, (__bf16*)k, (const __bf16*)cosv,
                           (const __bf16*)sinv, rows, Hq, Hk, (long)q_row_stride, (long)k_row_stride);
    else
        LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", dtype);
    LS_CHECK_LAUNCH("rope_apply_kernel");
    return LS_OK;
}

int ls_tree_positions(const int64_t* tree_mask, const int32_t* base, int b, int M, int N, int64_t* positions, void* stream) {
    if (!tree_mask || !positions || b < 1 || M < 1 || N < 1) LS_FAIL(LS_ERR_INVALID_ARG, "tree_positions args");
    const int rows = b * M;
    hipLaunchKernelGGL(tree_positions_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       tree_mask, base, M, N, positions, rows);
    LS_CHECK_LAUNCH("tree_positions_kernel");
    return LS_OK;
}

}  // extern "C"
