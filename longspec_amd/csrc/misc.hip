// Latency-bound operators of the decode round: RMSNorm, RoPE, tree positions and the
// accept/reject tree collapse.  KB-scale traffic each: single-pass kernels, vectorised
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
    const float f = (float)pos[r] * inv_freq[j];
    const float c = (float)cos((double)f) * scaling;
    const float s = (float)sin((double)f) * scaling;
    const typename E::T ce = E::from_f32(c), se = E::from_f32(s);
    cosv[(long)r * 128 + j] = ce;
    cosv[(long)r * 128 + 64 + j] = ce;
    sinv[(long)r * 128 + j] = se;
    sinv[(long)r * 128 + 64 + j] = se;
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

// ---- accept / reject tree collapse ------------------------------------------------------
// LlamaGlide.tree_verification (llama_glide.py:1128-1175).  One workgroup per batch row.
constexpr int MAXF = 1024;
__global__ __launch_bounds__(256) void tree_collapse_kernel(
    const int64_t* __restrict__ all_spec, const int64_t* __restrict__ all_pred, const int64_t* __restrict__ tree_mask,
    const int32_t* __restrict__ cache_lens, int Fn, int non_leaf_len, int max_acc, int64_t* __restrict__ acc_ids,
    int64_t* __restrict__ acc_num, int32_t* __restrict__ double_input, int64_t* __restrict__ index_mapping, char* k_cache,
    char* v_cache, long kc_sb_bytes, long kc_ss_bytes, int row_bytes) {
    __shared__ int father[MAXF];
    __shared__ unsigned char verify[MAXF];
    __shared__ int s_last;
    __shared__ int s_count;
    __shared__ int s_map[MAXF];
    const int z = blockIdx.x, tid = threadIdx.x;
    const int64_t* spec = all_spec + (long)z * Fn;
    const int64_t* pred = all_pred + (long)z * Fn;
    const int64_t* mask = tree_mask + (long)z * Fn * Fn;
    if (tid == 0) {
        s_last = 0;
        s_count = 0;
    }
    // father[r] = argmax_c((mask - I)[r,c] * c): the largest c != r with mask[r,c] != 0, else 0 (:1136)
    for (int r = tid; r < Fn; r += 256) {
        long best = 0;
        int bi = 0;
        for (int c = 0; c < Fn; ++c) {
            const long v = (mask[(long)r * Fn + c] - (r == c ? 1 : 0)) * (long)c;
            if (v > best) {
                best = v;
                bi = c;
            }
        }
        father[r] = bi;
    }
    __syncthreads();
    for (int r = tid; r < Fn; r += 256) verify[r] = (r == 0) || (pred[father[r]] == spec[r]);   // :1138-1139
    __syncthreads();
    // final[r] = sum_c mask[r,c]*verify[c] == sum_c mask[r,c]; last = argmax_r(final[r]*r)   (:1140-1144)
    for (int r = tid; r < Fn; r += 256) {
        long a = 0, t = 0;
        for (int c = 0; c < Fn; ++c) {
            const long mv = mask[(long)r * Fn + c];
            a += mv * (long)verify[c];
            t += mv;
        }
        if (a == t && r > 0) atomicMax(&s_last, r);
    }
    __syncthreads();
    const int last = s_last;
    // selected columns of mask[last] in ascending order (:1147-1154)
    if (tid < 64) {
        int base = 0;
        for (int c0 = 0; c0 < Fn; c0 += 64) {
            const int c = c0 + tid;
            const bool sel = (c < Fn) && (mask[(long)last * Fn + c] != 0);
            const unsigned long long bal = __ballot(sel);
            if (sel) {
                const int rank = base + __popcll(bal & ((1ull << tid) - 1ull));
                s_map[rank] = c;
            }
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
    for (int j = tid; j < max_acc; j += 256) {
        const int src = j < n_acc ? s_map[j] : -1;
        index_mapping[(long)z * max_acc + j] = src;
        acc_ids[(long)z * max_acc + j] = src >= 0 ? pred[src] : 0;       // :1155
    }
    // move the last target layer's KV rows cache_lens + map[j] -> cache_lens + j   (:1159-1173)
    if (k_cache == nullptr) return;
    const long L = cache_lens[z];
    const int chunks_per_row = row_bytes / 16;
    const int n_move = min(n_acc, max_acc);
    const int total = n_move * chunks_per_row;
    constexpr int MAXC = 8;                                 // 16-byte chunks per thread and tensor in flight
    for (int c0 = 0; c0 < total; c0 += 256 * MAXC) {
        uint4 kb[MAXC], vb[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int idx = c0 + i * 256 + tid;
            if (idx < total) {
                const int j = idx / chunks_per_row, ch = idx % chunks_per_row;
                const long src = (long)z * kc_sb_bytes + (L + s_map[j]) * kc_ss_bytes + (long)ch * 16;
                kb[i] = *reinterpret_cast<const uint4*>(k_cache + src);
                vb[i] = *reinterpret_cast<const uint4*>(v_cache + src);
            }
        }
        // rows are moved towards lower indices (map[j] >= j): a chunk group never overwrites a
        // source row of a LATER group only if every read of this group finished first
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int idx = c0 + i * 256 + tid;
            if (idx < total) {
                const int j = idx / chunks_per_row, ch = idx % chunks_per_row;
                const long dst = (long)z * kc_sb_bytes + (L + j) * kc_ss_bytes + (long)ch * 16;
                *reinterpret_cast<uint4*>(k_cache + dst) = kb[i];
                *reinterpret_cast<uint4*>(v_cache + dst) = vb[i];
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int ls_version(void) { return 100; }
const char* ls_last_error(void) { return g_err; }

int ls_rmsnorm_fwd(const void* x, const void* residual, const void* weight, void* y, void* sum_out, int rows, int hidden,
                   float eps, int dtype, void* stream) {
    if (!x || !weight || !y || rows < 1 || hidden < 8 || (hidden & 7)) LS_FAIL(LS_ERR_INVALID_ARG, "rmsnorm args");
    const size_t lds = ((hidden * 2 + 15) & ~15) + 16;
    if (lds > 64 * 1024) LS_FAIL(LS_ERR_UNSUPPORTED, "hidden %d too large", hidden);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LS_F16)
        hipLaunchKernelGGL(rmsnorm_kernel<ElemF16>, dim3(rows), dim3(256), lds, s, (const _Float16*)x,
                           (const _Float16*)residual, (const _Float16*)weight, (_Float16*)y, (_Float16*)sum_out, hidden, eps);
    else if (dtype == LS_BF16)
        hipLaunchKernelGGL(rmsnorm_kernel<ElemBF16>, dim3(rows), dim3(256), lds, s, (const __bf16*)x,
                           (const __bf16*)residual, (const __bf16*)weight, (__bf16*)y, (__bf16*)sum_out, hidden, eps);
    else
        LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", dtype);
    LS_CHECK_LAUNCH("rmsnorm_kernel");
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

int ls_tree_collapse(const int64_t* all_spec, const int64_t* all_llm_pred, const int64_t* tree_mask,
                     const int32_t* cache_lens, int b, int F, int non_leaf_len, int max_acc, int64_t* acc_ids,
                     int64_t* acc_num, int32_t* double_input, int64_t* index_mapping, void* k_cache, void* v_cache,
                     int64_t kc_stride_b, int64_t kc_stride_s, int row_elems, int dtype, void* stream) {
    if (!all_spec || !all_llm_pred || !tree_mask || !acc_ids || !acc_num || !double_input || !index_mapping)
        LS_FAIL(LS_ERR_INVALID_ARG, "tree_collapse: null pointer");
    if (b < 1 || F < 1 || F > MAXF || max_acc < 1 || max_acc > F) LS_FAIL(LS_ERR_INVALID_ARG, "tree_collapse: F=%d max_acc=%d", F, max_acc);
    if (k_cache) {
        if (!v_cache || !cache_lens || row_elems < 8 || (row_elems & 7) || (kc_stride_s & 7))
            LS_FAIL(LS_ERR_INVALID_ARG, "tree_collapse: KV move args");
        if (dtype != LS_F16 && dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", dtype);
    }
    hipLaunchKernelGGL(tree_collapse_kernel, dim3(b), dim3(256), 0, static_cast<hipStream_t>(stream), all_spec,
                       all_llm_pred, tree_mask, cache_lens, F, non_leaf_len, max_acc, acc_ids, acc_num, double_input,
                       index_mapping, (char*)k_cache, (char*)v_cache, (long)kc_stride_b * 2, (long)kc_stride_s * 2,
                       row_elems * 2);
    LS_CHECK_LAUNCH("tree_collapse_kernel");
    return LS_OK;
}

}  // extern "C"
