/*
 * C restatement of the hybrid tree-verification attention (the dominant operator of the
 * draft-then-verify round) -- TEST INFRASTRUCTURE ONLY, like everything under oracle/: it is
 * the checker and the `cpu_baseline` ("kind": "port") leg of bench.py, never a product path.
 *
 * Follows oracle/ref_ops.py::target_verify_attention line by line, i.e.
 *   prefix      flash_attn_with_kvcache contract          longspec/test/llama.py:385
 *   scatter     K/V rows into the cache at cache_lens+i   longspec/test/llama.py:396-399
 *   tree part   LlamaAttention.tree_part_fwd              longspec/test/llama.py:401-420
 *   merge       prefix_o*w + current_out*(1-w) in fp16    longspec/test/llama.py:387
 * Pinned by tests/test_oracle_c.py against ref_ops (itself pinned against the reference's goldens).
 * OpenMP over query heads; plain C11, fp16 <-> fp32 by bit manipulation (round-to-nearest-even).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define D 128

static inline float h2f(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
    if (e == 0) {
        if (m == 0) { u = s; }
        else { int sh = 0; while (!(m & 0x400)) { m <<= 1; ++sh; } m &= 0x3ff; u = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) { u = s | 0x7f800000u | (m << 13); }
    else { u = s | ((e + 112) << 23) | (m << 13); }
    float f; memcpy(&f, &u, 4); return f;
}

static inline uint16_t f2h(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000; int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15; uint32_t m = u & 0x7fffff;
    if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(s | 0x7c00 | (m ? 0x200 : 0));
    if (e >= 31) return (uint16_t)(s | 0x7c00);
    if (e <= 0) {
        if (e < -10) return (uint16_t)s;
        m |= 0x800000; int sh = 14 - e; uint32_t r = m >> sh, rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (r & 1))) ++r;
        return (uint16_t)(s | r);
    }
    uint32_t r = (uint32_t)(e << 10) | (m >> 13), rem = m & 0x1fff;
    if (rem > 0x1000 || (rem == 0x1000 && (r & 1))) ++r;
    return (uint16_t)(s | r);
}
/* bfloat16 <-> fp32 (round-to-nearest-even, NaN kept quiet): what `tensor.to(torch.bfloat16)` does */
static inline float b2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t f2b(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
/* storage dtype of the call: 0 = fp16, 1 = bf16 (file-scope so that the loops below read like the fp16 original; set once per
 * call before the parallel region, read-only inside it) */
static int g_bf16 = 0;
static inline float ld(uint16_t h) { return g_bf16 ? b2f(h) : h2f(h); }
static inline uint16_t st(float f) { return g_bf16 ? f2b(f) : f2h(f); }
static inline float rh(float f) { return ld(st(f)); }   /* one rounding to the storage dtype */

/* q [R,H,D], k_new/v_new [R,Hkv,D], caches [S,Hkv,D] (row stride Hkv*D), tree_mask [R,R] int64 */
static int verify_attention(const uint16_t* q, const uint16_t* k_new, const uint16_t* v_new, uint16_t* k_cache,
                            uint16_t* v_cache, int L, const int64_t* tree_mask, int R, int H, int Hkv, int last_layer,
                            float scale, uint16_t* out) {
    const int g = H / Hkv;
    const size_t rs = (size_t)Hkv * D;
    /* scatter (llama.py:396-399) */
    for (int i = 0; i < R; ++i) {
        memcpy(k_cache + (size_t)(L + i) * rs, k_new + (size_t)i * rs, rs * 2);
        memcpy(v_cache + (size_t)(L + i) * rs, v_new + (size_t)i * rs, rs * 2);
    }
    int fail = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        const int hk = h / g;
        float* qf = (float*)malloc(sizeof(float) * R * D);
        float* s = (float*)malloc(sizeof(float) * (size_t)R * (L > R ? L : R));
        float* kf = (float*)malloc(sizeof(float) * 64 * D);
        float* acc = (float*)malloc(sizeof(float) * R * D);
        float* pre_o = (float*)malloc(sizeof(float) * R * D);
        float* pre_lse = (float*)malloc(sizeof(float) * R);
        float* lsum = (float*)malloc(sizeof(float) * R);
        if (!qf || !s || !kf || !acc || !pre_o || !pre_lse || !lsum) { fail = 1; goto done; }
        for (int r = 0; r < R; ++r)
            for (int d = 0; d < D; ++d) qf[r * D + d] = ld(q[((size_t)r * H + h) * D + d]);
        /* ---- prefix: fp32 scores, fp32 statistics, P -> fp16 before P.V, one division (ref_ops._attend) */
        for (int j0 = 0; j0 < L; j0 += 64) {
            const int nj = L - j0 < 64 ? L - j0 : 64;
            for (int j = 0; j < nj; ++j)
                for (int d = 0; d < D; ++d) kf[j * D + d] = ld(k_cache[(size_t)(j0 + j) * rs + hk * D + d]);
            for (int r = 0; r < R; ++r)
                for (int j = 0; j < nj; ++j) {
                    float a = 0.f;
                    for (int d = 0; d < D; ++d) a += qf[r * D + d] * kf[j * D + d];
                    s[(size_t)r * L + j0 + j] = a * scale;
                }
        }
        for (int r = 0; r < R; ++r) {
            float m = -INFINITY, l = 0.f;
            for (int j = 0; j < L; ++j) m = fmaxf(m, s[(size_t)r * L + j]);
            for (int j = 0; j < L; ++j) { float p = expf(s[(size_t)r * L + j] - m); l += p; s[(size_t)r * L + j] = rh(p); }
            pre_lse[r] = L > 0 ? m + logf(l) : -INFINITY;
            for (int d = 0; d < D; ++d) acc[r * D + d] = 0.f;
            lsum[r] = l;
        }
        for (int j0 = 0; j0 < L; j0 += 64) {
            const int nj = L - j0 < 64 ? L - j0 : 64;
            for (int j = 0; j < nj; ++j)
                for (int d = 0; d < D; ++d) kf[j * D + d] = ld(v_cache[(size_t)(j0 + j) * rs + hk * D + d]);
            for (int r = 0; r < R; ++r)
                for (int j = 0; j < nj; ++j) {
                    const float p = s[(size_t)r * L + j0 + j];
                    for (int d = 0; d < D; ++d) acc[r * D + d] += p * kf[j * D + d];
                }
        }
        for (int r = 0; r < R; ++r) {
            const float l = lsum[r];
            for (int d = 0; d < D; ++d) pre_o[r * D + d] = L > 0 ? rh(acc[r * D + d] / l) : 0.f;
        }
        /* ---- tree part (llama.py:401-420): fp16 QK^T, scale before/after (G1), fp32 softmax, P -> fp16 */
        for (int r = 0; r < R; ++r) {
            float sc[128];
            float m = -INFINITY;
            for (int j = 0; j < R; ++j) {
                float a = 0.f;
                for (int d = 0; d < D; ++d) {
                    const float qv = last_layer ? rh(qf[r * D + d] * scale) : qf[r * D + d];
                    a += qv * ld(k_new[((size_t)j * Hkv + hk) * D + d]);
                }
                a = rh(a);
                if (!last_layer) a = rh(a * scale);
                sc[j] = tree_mask[(size_t)r * R + j] ? a : -INFINITY;
                m = fmaxf(m, sc[j]);
            }
            float l = 0.f;
            for (int j = 0; j < R; ++j) { sc[j] = expf(sc[j] - m); l += sc[j]; }
            const float cur_lse = m + logf(l);
            float cur[D];
            for (int d = 0; d < D; ++d) cur[d] = 0.f;
            for (int j = 0; j < R; ++j) {
                const float p = rh(sc[j] / l);
                for (int d = 0; d < D; ++d) cur[d] += p * ld(v_new[((size_t)j * Hkv + hk) * D + d]);
            }
            const float w = rh(1.0f / (1.0f + expf(-(pre_lse[r] - cur_lse))));
            const float omw = rh(1.0f - w);
            for (int d = 0; d < D; ++d) {
                const float a = rh(pre_o[r * D + d] * w), b = rh(rh(cur[d]) * omw);
                out[((size_t)r * H + h) * D + d] = st(a + b);
            }
        }
    done:
        free(qf); free(s); free(kf); free(acc); free(pre_o); free(pre_lse); free(lsum);
    }
    return fail ? -1 : 0;
}

int oracle_verify_attention_f16(const uint16_t* q, const uint16_t* k_new, const uint16_t* v_new, uint16_t* k_cache,
                                uint16_t* v_cache, int L, const int64_t* tree_mask, int R, int H, int Hkv, int last_layer,
                                float scale, uint16_t* out) {
    g_bf16 = 0;
    return verify_attention(q, k_new, v_new, k_cache, v_cache, L, tree_mask, R, H, Hkv, last_layer, scale, out);
}

/* the same restatement on bfloat16 storage (how inference_qwq.py runs QwQ-32B: every rounding point of the reference rounds to
 * the activation dtype, llama.py:387,406-420 with dtype = bfloat16) */
int oracle_verify_attention_bf16(const uint16_t* q, const uint16_t* k_new, const uint16_t* v_new, uint16_t* k_cache,
                                 uint16_t* v_cache, int L, const int64_t* tree_mask, int R, int H, int Hkv, int last_layer,
                                 float scale, uint16_t* out) {
    g_bf16 = 1;
    const int rc = verify_attention(q, k_new, v_new, k_cache, v_cache, L, tree_mask, R, H, Hkv, last_layer, scale, out);
    g_bf16 = 0;
    return rc;
}
