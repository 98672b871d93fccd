// Hybrid tree-verification / draft-step attention for gfx950 (CDNA4).
//
// Stage 1  attn_partial_kernel : split-KV "flash-decoding" over the long prefix KV
//          (one workgroup per (kv head, key split)), plus one workgroup per kv head
//          for the "new key block" (the 74 tree tokens of a verify pass / the tree
//          levels of a draft step / appended decode tokens) under a bit mask.
// Stage 2  attn_finish_kernel  : log-sum-exp combine of the split partials and the
//          reference-order merge with the new-block part.
//
// Reference seams replaced (paths relative to the reference root):
//   flash_attn_with_kvcache       longspec/test/llama.py:324,385  llama_glide.py:261,265,297,300
//   LlamaAttention.tree_part_fwd  longspec/test/llama.py:394-421   (+ merge :387)
//   triton_tree_attn._fwd_kernel  longspec/test/triton_tree_attn.py:115-251 (+ merge llama_glide.py:302)
//
// MI355X mapping.  All g = H/Hkv query heads of a kv head and all sq query rows are
// packed into one M = g*sq row block that shares every K/V byte read from HBM (GQA-4,
// 74 rows: 296 flop per KV byte, i.e. right at the MFMA/HBM ridge).  A workgroup is
// RB x KS waves (<= 8: two per SIMD, 256 registers each): RB row blocks times KS key slices (small
// row counts).  A row block is QT tiles of 16 rows; the 296 rows (19 tiles -> 20) of a Llama-3
// verify pass are dealt 3,3,3,3,2,2,2,2 so that every SIMD (waves w and w+4) carries 5 tiles --
// balanced matrix work, and the second wave of a SIMD covers the LDS / MFMA / VALU latencies of
// the first.  Per 32-key block a wave computes
//     S^T[key][row] = K . Q^T          (mfma_f32_16x16x32: A = K fragment,  B = Q^T fragment)
//     O^T[d][row]  += V^T . P^T        (A = V^T via ds_read_b64_tr_b16,     B = P^T = S^T's own layout)
// Both products are "transposed" so that a lane always owns ONE query row (lane&15):
// the soft-max running max / sum / rescale are per-lane scalars (two xor-shuffles per
// row for the max), P feeds the second MFMA straight from registers, and nothing but
// K/V tiles ever goes through LDS.  K/V tiles go HBM -> LDS directly (global_load_lds,
// 16 bytes per lane, 256 contiguous bytes per key and kv head, no staging registers) into a
// ring of up to 4 tiles: with HBM round trips of ~3 us under load a single tile in flight
// caps a CU at ~10 GB/s, so tiles t+1..t+3 are in flight while tile t is multiplied
// (counted s_waitcnt vmcnt + raw s_barrier, one barrier per tile).  The LDS image is
// XOR-swizzled (applied on the per-lane SOURCE address, the LDS destination of the DMA being
// lane-linear) so that both the ds_read_b128 K-fragment reads and the transposing V reads are
// bank-conflict free.
#include <stdlib.h>
#include <type_traits>

#include "ls_common.h"

namespace {

// Diagnostic (ls_attn_redo_count; tools/bench_attn.py --sink / --score-scale): how many (key split, kv head) workgroups had to
// redo their split because a soft-max numerator left the fp16 range of the fixed reference.  Bumped on that rare path only.
__device__ unsigned int g_attn_redo_count = 0;


constexpr int D = LS_HEAD_DIM;      // 128
constexpr int ROWB = D * 2;         // bytes per key row (fp16/bf16)
constexpr float LOG2E = 1.4426950408889634f;
constexpr int MAX_THREADS = 512;    // up to 8 waves per workgroup: 2 per SIMD, 256 registers each

struct AttnK {
    const void* q;
    const void* k_cache;
    const void* v_cache;
    void* k_cache_w;
    void* v_cache_w;
    const void* k_new;
    const void* v_new;
    const int32_t* cache_seqlens;
    const uint32_t* mask_bits;
    float* parts_o;    // [n_parts][b][sq][H][D]
    float* parts_lse;  // [n_parts][b][H][sq]
    float* new_o;      // [b][sq][H][D]
    float* new_lse;    // [b][H][sq]
    int b, sq, H, Hkv, g, M;
    int has_new, new_mode, n_new, n_new_cached, mask_words, scatter_new, prescale_q;
    int causal, window_left, n_app;
    int n_splits, row_chunks, rows_per_chunk;
    // workgroup shape (host-chosen): RB row blocks x KS key slices waves; a tile = `tile` keys
    // (KS * bpw * 32), `nstages` tiles of LDS ring, DMA by the first `nd` waves, `pp` pieces each
    int RB, KS, tile, bpw, nstages, nd, pp;
    // row blocks 0..rbA-1 have qtA tiles of 16 rows, the others qtB
    int rbA, qtA, qtB;
    int pp_extra;       // ping-pong kernel: row tiles - 16 (the first pp_extra waves carry 3 tiles, the others 2)
    float scale;
    long q_sb, q_ss, q_sh;
    long kc_sb, kc_ss, kc_sh;
    long kn_sb, kn_ss, kn_sh;
};

template <typename E, int QT>
struct WaveAcc {
    f32x4 acc[8][QT];   // O^T[d-tile][q-tile]: lane holds d = dt*16 + g4*4 + reg for row l15
    float m[QT];        // running max (raw score units)
    float l[QT];        // running sum of exp (this lane's keys only)
};

// ---- LDS tile layout -------------------------------------------------------------
// K: row-major [row][128], 16-byte slot s stored at slot s ^ (row & 15)
// V: row-major [key][128], 16-byte slot s stored at slot s ^ ((key & 7) << 1)
// Fragment addresses (rows 16-aligned => (row & 15) == l15), written so that the per-lane part
// is ONE register per operand and the k4 / dt dependence is an XOR on the fly:
//   K/Q fragment k4 :  kb + ((k4 ^ (l15 >> 2)) << 6),   kb = l15*256 + ((g4 ^ (l15 & 3)) << 4)
//   V^T fragment dt :  vb + ((dt ^ kk) << 5),           vb = key*256 + (((l15 >> 1) & 1) << 4) + ((l15 & 1) << 3)
//                      key = g4*4 + (l15 >> 2), kk = key & 7
struct LaneTbl {
    int kb, kx;      // K: base, xor key (l15 >> 2)
    int vb, vx;      // V: base, xor key (key & 7)
};

__device__ __forceinline__ LaneTbl make_lane_tbl(int l15, int g4) {
    LaneTbl t;
    t.kb = l15 * ROWB + ((g4 ^ (l15 & 3)) << 4);
    t.kx = l15 >> 2;
    const int key = g4 * 4 + (l15 >> 2);
    t.vb = key * ROWB + (((l15 >> 1) & 1) << 4) + ((l15 & 1) << 3);
    t.vx = key & 7;
    return t;
}

template <typename V8>
__device__ __forceinline__ V8 lds_read16(unsigned addr) {
    typedef __attribute__((address_space(3))) V8 lds_v8;
    return *(lds_v8*)(uintptr_t)addr;
}

// S^T for one 32-key block: s[kt][qt], kt = 16-key tile.  kbase = LDS byte address of the
// block's first key row.
template <typename E, int QT>
__device__ __forceinline__ void qk_block(f32x4 (&s)[2][QT], const typename E::V8 (&qf)[QT][4], const LaneTbl& tb,
                                         unsigned kbase) {
    // keep the 4 fragment addresses out of loop-invariant hoisting (they would be precomputed per
    // ring slot and spilled): the xor key is made opaque once per block
    int kx = tb.kx;
    asm volatile("" : "+v"(kx));
    const unsigned kb = kbase + tb.kb;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        const unsigned ka = kb + ((k4 ^ kx) << 6);
        typename E::V8 kf[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) kf[kt] = lds_read16<typename E::V8>(ka + kt * 16 * ROWB);
        __builtin_amdgcn_s_setprio(1);          // keep the matrix pipe fed while the SIMD partner issues LDS / VALU work
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) s[kt][qt] = E::mfma(kf[kt], qf[qt][k4], s[kt][qt]);
        __builtin_amdgcn_s_setprio(0);
    }
}

// O^T += V^T . P^T for one 32-key block.  pf[qt] = 8 probabilities of row l15:
// elements 0..3 = keys g4*4 + e, elements 4..7 = keys 16 + g4*4 + e of the block
// (exactly the S^T accumulator layout of the two 16-key tiles).  The A operand V^T
// [16 d x 32 keys] comes from two transposing LDS reads: within a 16-lane group, lane p
// supplies the address of V[key0 + p/4][d0 + (p%4)*4 .. +3] and receives V[key0 + j][d0 + p],
// j = 0..3 (ds_read_b64_tr_b16).
template <typename E, int QT>
__device__ __forceinline__ void pv_block(WaveAcc<E, QT>& w, const typename E::V8 (&pf)[QT], const LaneTbl& tb,
                                         unsigned vbase) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    int vx = tb.vx;
    asm volatile("" : "+v"(vx));           // see qk_block
    const unsigned vb = vbase + tb.vb;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
        const unsigned va = vb + ((dt ^ vx) << 5);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)va);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(va + 16 * ROWB));
        union {
            struct { s16x4 a, b; } s;
            typename E::V8 v;
        } u;
        u.s.a = lo;
        u.s.b = hi;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) w.acc[dt][qt] = E::mfma(u.v, pf[qt], w.acc[dt][qt]);
    }
}

// Soft-max update (base-2 exponentials) for one 32-key block whose masked scores are already
// -inf, then P.V.
//   SAFE = true : textbook online soft-max -- block row-max, running max m, rescale of O and l
//                 whenever a row's max grows.
//   SAFE = false: the running max is used as a FIXED reference: p = 2^(s*c - m*c) without looking
//                 at the block's own max, so the accumulator registers are touched by MFMAs only.
//                 Exact as long as no p overflows fp16 (s - m < ~11 in natural-log units, i.e. a
//                 key scoring e^11 above everything seen so far); the largest p is tracked in
//                 `pmax` and the caller re-runs the split with SAFE = true if it ever gets near the
//                 fp16 range.  (lse = m*scale + ln(l) holds for any reference m.)
// Round 4: the fixed-reference form raises its reference LAZILY.  The eight numerators of a lane are summed in fp32 before they
// are rounded to 16 bits; a sum beyond 2^12 (a key 8+ nats above everything the split has seen: a retrieval spike, a heavy
// tail) triggers, wave-uniformly and in the same block, the textbook step -- row maximum, rescale of O and l, new reference --
// and the block's numerators are recomputed against it.  Until round 3 such a block only raised a flag and the whole split
// ran again in textbook form (1.75-2x per call on N(0, 4^2) logits or a few +14-nat keys: profiles/r4_redo_general_kernel.jsonl);
// a call whose scores stay within 8 nats of their running reference executes exactly the instructions it did.
constexpr float LAZY_RAISE_SUM = 4096.f;
template <typename E, int QT>
__device__ __forceinline__ void raise_reference(WaveAcc<E, QT>& w, const f32x4 (&s)[2][QT], int qt, float c) {
    float mx = fmaxf(fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), fmaxf(s[0][qt][2], s[0][qt][3])),
                     fmaxf(fmaxf(s[1][qt][0], s[1][qt][1]), fmaxf(s[1][qt][2], s[1][qt][3])));
    mx = wave_xor_max_16_32(mx);
    const float m_new = fmaxf(w.m[qt], mx);
    if (m_new > w.m[qt]) {                          // (per lane = per row: rows below their reference keep every bit)
        const float alpha = __builtin_amdgcn_exp2f((w.m[qt] - m_new) * c);
        w.l[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) w.acc[dt][qt] *= alpha;
        w.m[qt] = m_new;
    }
}

template <typename E, int QT, bool SAFE>
__device__ __forceinline__ void online_block(WaveAcc<E, QT>& w, const f32x4 (&s)[2][QT], float c, const LaneTbl& tb,
                                             unsigned vbase, float& pmax) {
    typename E::V8 pf[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float mc;
        if (SAFE) {
            float mx = fmaxf(fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), fmaxf(s[0][qt][2], s[0][qt][3])),
                             fmaxf(fmaxf(s[1][qt][0], s[1][qt][1]), fmaxf(s[1][qt][2], s[1][qt][3])));
            mx = wave_xor_max_16_32(mx);
            const float m_new = fmaxf(w.m[qt], mx);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;     // row with no visible key so far
            if (__any(m_new > w.m[qt])) {
                const float alpha = __builtin_amdgcn_exp2f((w.m[qt] - m_safe) * c);
                w.l[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) w.acc[dt][qt] *= alpha;
                w.m[qt] = m_new;
            }
            mc = m_safe * c;
        } else {
            mc = w.m[qt] * c;
        }
        float pe[8];
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pe[kt * 4 + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][e], c, -mc));
                ps += pe[kt * 4 + e];
            }
        if (!SAFE && __any(!(ps <= LAZY_RAISE_SUM))) {      // rare (also catches a non-finite sum)
            raise_reference<E, QT>(w, s, qt, c);
            mc = w.m[qt] * c;
            ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pe[kt * 4 + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][e], c, -mc));
                    ps += pe[kt * 4 + e];
                }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[qt][e] = E::from_f32(pe[e]);
        w.l[qt] += ps;
        if (!SAFE) pmax = fmaxf(pmax, ps);   // (stays far below the redo threshold now: kept as the safety net it was)
    }
    pv_block<E, QT>(w, pf, tb, vbase);
}

// Fixed-reference soft-max of one 32-key block (SAFE = false form of online_block) WITHOUT the P.V
// product: returns the probabilities as the B operand of the next P.V so that the caller can issue
// them under the MFMAs of the previous block's P.V (software pipeline of the interior loop).
template <typename E, int QT>
__device__ __forceinline__ void softmax_fast(WaveAcc<E, QT>& w, const f32x4 (&s)[2][QT], float c, typename E::V8 (&pf)[QT],
                                             float& pmax) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float mc = w.m[qt] * c;
        float pe[8];
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pe[kt * 4 + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][e], c, -mc));
                ps += pe[kt * 4 + e];
            }
        if (__any(!(ps <= LAZY_RAISE_SUM))) {       // rare: see raise_reference (the caller's P.V of the PREVIOUS block was issued first)
            raise_reference<E, QT>(w, s, qt, c);
            mc = w.m[qt] * c;
            ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pe[kt * 4 + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][e], c, -mc));
                    ps += pe[kt * 4 + e];
                }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[qt][e] = E::from_f32(pe[e]);
        w.l[qt] += ps;
        pmax = fmaxf(pmax, ps);
    }
}

// ---- HBM -> LDS tile DMA -----------------------------------------------------------------
// One wave-instruction ("piece") moves 64 x 16 B = 4 key rows of K or of V.  LDS slot `pos` of
// row `key` receives global chunk pos ^ swz(key) (an involution: the same XOR is used on reads).
// A tile of `tile` keys is tile/4 K pieces + tile/4 V pieces; DMA wave w (< nd) issues the key
// groups w, w + nd, ... (pp/2 of them, K and V each).
typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lds_v;

// One 16-byte-per-lane LDS DMA: lane i's 16 bytes at `src` land at LDS byte address lds_base + 16*i.  Issued as
// inline assembly on purpose: for the compiler's own global_load_lds builtin LLVM tracks "an LDS DMA is pending"
// and, having no alias information, puts s_waitcnt vmcnt(0) in front of every later LDS read it recognises (the
// ds_read_b64_tr_b16 of the P.V product) -- which drains the whole DMA look-ahead once per 32-key block (measured:
// ~2 TB/s).  Through asm the pending DMAs are invisible to it; the landing of a tile is awaited explicitly with
// counted s_waitcnt vmcnt(N) + s_barrier.  (Its vmcnt bookkeeping for ordinary loads can only become more
// conservative: the counter is in order and these ops are simply not counted.)
__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_char*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(base) : "memory", "m0");
}

// Cache policy of the streamed prefix K/V (every byte is used once per call by one workgroup): -DLS_KV_NT=1 requests it
// non-temporally (MI355X_MICROARCH "nt-weights"; round 4 A/B on one box: verification call 0.4114 -> 0.4175 of the HBM roofline,
// round 12.28 -> 12.15 ms); 0 = default policy.
#ifndef LS_KV_NT
#define LS_KV_NT 1
#endif
#if LS_KV_NT
#define LS_KV_POLICY " nt"
#else
#define LS_KV_POLICY ""
#endif
// one K and one V piece (4 keys each) of the fast DMA path: wave-uniform 64-bit base, 32-bit per-lane offset
__device__ __forceinline__ void dma16_s(const char* base_uniform, unsigned lane_off, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" LS_KV_POLICY
                 :
                 : "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform)
                 : "memory", "m0");
}

// A wave must not end with LDS DMA of its own still in flight (look-ahead tiles it issued and never needed): the
// workgroup's LDS is handed to the next workgroup on the CU as soon as its waves are gone, and a late global_load_lds
// then lands in THAT workgroup's tiles.  One dispatch round (<= 256 workgroups: every decode call) never shows it; a
// batch of 64 identical elements did (tools/bench_prefill.py history).
__device__ __forceinline__ void drain_lds_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename F>
__device__ __forceinline__ void tile_dma(char* ldsK, char* ldsV, int wave, int lane, int nd, int ngroups, F&& row_ptr) {
    const int kq = lane >> 4, pos = lane & 15;
    for (int grp = wave; grp < ngroups; grp += nd) {
        const int key = grp * 4 + kq;
        const char* kp;
        const char* vp;
        row_ptr(key, kp, vp);
        dma16(kp + ((pos ^ (key & 15)) << 4), ldsK + grp * 1024);
        dma16(vp + ((pos ^ ((key & 7) << 1)) << 4), ldsV + grp * 1024);
    }
}

// wait until at most n of this wave's vector-memory operations (= DMA pieces) are outstanding
__device__ __forceinline__ void wait_vmcnt(int n) {
    if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- per-wave context -------------------------------------------------------------------
template <typename E, int QT>
struct Ctx {
    int tid, lane, wave, rb, ks, l15, g4, bi, kvh, chunk, L, sk, row0;
    float c;
    int rrow[QT];
    unsigned smem_a;
    LaneTbl tb;
    const char* kc_base;
    const char* vc_base;
    long kc_row;
};

template <typename E, int QT>
__device__ __forceinline__ void ctx_init(Ctx<E, QT>& x, const AttnK& p, char* smem) {
    x.tid = threadIdx.x;
    x.lane = x.tid & 63;
    x.wave = __builtin_amdgcn_readfirstlane(x.tid >> 6);
    x.rb = x.wave / p.KS;
    x.ks = x.wave % p.KS;
    x.l15 = x.lane & 15;
    x.g4 = x.lane >> 4;
    x.bi = blockIdx.z;
    x.kvh = blockIdx.y % p.Hkv;
    x.chunk = blockIdx.y / p.Hkv;
    x.L = p.cache_seqlens[x.bi];
    x.c = p.scale * LOG2E;
    x.sk = x.L + p.n_app;
    // rows of this wave: lane l15 of q-tile qt owns row m = row0 + qt*16 + l15
    x.row0 = x.chunk * p.rows_per_chunk + (x.rb < p.rbA ? x.rb * p.qtA : p.rbA * p.qtA + (x.rb - p.rbA) * p.qtB) * 16;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int m = x.row0 + qt * 16 + x.l15;
        x.rrow[qt] = m < p.M ? m % p.sq : 0;     // 0 for padding rows: computed, never stored
    }
    typedef __attribute__((address_space(3))) char lds_char;
    x.smem_a = (unsigned)(uintptr_t)(lds_char*)smem;          // LDS byte address of the carve
    x.tb = make_lane_tbl(x.l15, x.g4);
    x.kc_base = reinterpret_cast<const char*>(p.k_cache) + ((long)x.bi * p.kc_sb + (long)x.kvh * p.kc_sh) * 2;
    x.vc_base = reinterpret_cast<const char*>(p.v_cache) + ((long)x.bi * p.kc_sb + (long)x.kvh * p.kc_sh) * 2;
    x.kc_row = p.kc_ss * 2;
}

// Q^T fragments (B operand of the first product), in registers.
template <typename E, int QT>
__device__ __forceinline__ void load_q(const Ctx<E, QT>& x, const AttnK& p, bool prescale, typename E::V8 (&qf)[QT][4]) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int m = x.row0 + qt * 16 + x.l15;
        const int head = x.kvh * p.g + (m < p.M ? m / p.sq : 0);
        const typename E::T* qp = reinterpret_cast<const typename E::T*>(p.q) + (long)x.bi * p.q_sb +
                                  (long)x.rrow[qt] * p.q_ss + (long)head * p.q_sh + x.g4 * 8;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            typename E::V8 v = *reinterpret_cast<const typename E::V8*>(qp + k4 * 32);
            if (prescale) {   // `query_states * self.softmax_scale` in the activation dtype (llama.py:407)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = E::from_f32(E::to_f32(v[e]) * p.scale);
            }
            qf[qt][k4] = v;
        }
    }
}

template <typename E, int QT>
__device__ __forceinline__ void acc_init(WaveAcc<E, QT>& w) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        w.m[qt] = -INFINITY;
        w.l[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) w.acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ================= prefix split: tiles [t_begin, t_end) of `tile` keys =====================
template <typename E, int QT>
__device__ __forceinline__ void prefix_path(const AttnK& p, char* smem, int split) {
    using C = Ctx<E, QT>;
    C x;
    ctx_init(x, p, smem);
    const int L = x.L, sk = x.sk, l15 = x.l15, g4 = x.g4;
    const int TILE = p.tile, S = p.nstages;
    const int STAGE = 2 * TILE * ROWB;            // bytes of one (K,V) stage
    // visible prefix key range of row r (flash-attn bottom-right alignment):
    //   lo(r) = max(0, r + sk - sq - window_left), hi(r) = min(L, r + sk - sq + 1) if causal else L
    int lo_min = 0, lo_max = 0, hi_min = L, hi_max = L;
    if (p.window_left >= 0) {
        lo_min = max(0, sk - p.sq - p.window_left);
        lo_max = max(0, sk - 1 - p.window_left);
    }
    if (p.causal) {
        hi_min = max(0, min(L, sk - p.sq + 1));
        hi_max = max(0, min(L, sk));
    }
    typename E::V8 qf[QT][4];
    load_q<E, QT>(x, p, false, qf);
    WaveAcc<E, QT> w;

    const int t0 = lo_min / TILE;
    const int t1 = (hi_max + TILE - 1) / TILE;
    const int tps = (max(t1 - t0, 0) + p.n_splits - 1) / p.n_splits;
    const int t_begin = t0 + split * tps;
    const int t_end = min(t_begin + tps, t1);
    const int last_key = hi_max - 1;
    const bool dma_wave = x.wave < p.nd;
    auto dma = [&](int tile) {              // tile -> ring slot (tile - t_begin) % S
        if (!dma_wave) return;
        char* bK = smem + ((tile - t_begin) % S) * STAGE;
        tile_dma(bK, bK + TILE * ROWB, x.wave, x.lane, p.nd, TILE / 4, [&](int key, const char*& kp, const char*& vp) {
            const long ka = min(tile * TILE + key, last_key);   // tail rows: re-read the last valid key (masked below)
            kp = x.kc_base + ka * x.kc_row;
            vp = x.vc_base + ka * x.kc_row;
        });
    };
    // Three tile ranges share one DMA pipeline:  [t_begin, tA) and [tB, t_end) touch a range edge
    // (window / causal / tail of the cache) or prime the running max and run the textbook online
    // soft-max with masks; [tA, tB) are interior tiles and run the fixed-reference form whose loop
    // body is branch-free: wait, barrier, DMA issue, bpw x (QK^T, exp2, P.V).
    // attempt 1 (only if some p came close to the fp16 range): textbook form everywhere.
    int* redo_flag = reinterpret_cast<int*>(smem + S * STAGE);
    float pmax = 0.f;
    // DMA schedule: LA = max(S-2, 1) tiles ahead.  One ring slot stays untouched behind the tile being
    // multiplied because the interior loop defers the P.V of a tile's last block into the next iteration.
    const int LA = S > 2 ? S - 2 : 1;
    auto tile_head = [&](int t) -> unsigned {
        // tile t landed once at most the pieces of the younger tiles in flight are outstanding
        const int younger = min(LA - 1, t_end - 1 - t);
        wait_vmcnt(dma_wave ? younger * p.pp : 0);
        __builtin_amdgcn_s_barrier();          // every wave's pieces of tile t landed; iteration t-1 is over everywhere
        if (t + LA < t_end) dma(t + LA);
        return x.smem_a + ((t - t_begin) % S) * STAGE;      // K tile; V tile follows at + TILE*ROWB
    };
    auto run_safe = [&](int t_from, int t_to) {
        for (int t = t_from; t < t_to; ++t) {
            const unsigned kb_a = tile_head(t);
#pragma unroll 1
            for (int blk = 0; blk < p.bpw; ++blk) {
                const int krow0 = (x.ks * p.bpw + blk) * 32;
                const int ka0 = t * TILE + krow0;
                if (ka0 >= hi_max || ka0 + 32 <= lo_min) continue;   // wave-uniform
                f32x4 s[2][QT];
                qk_block<E, QT>(s, qf, x.tb, kb_a + krow0 * ROWB);
                if (!(ka0 >= lo_max && ka0 + 32 <= hi_min)) {        // block touches a range edge
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const int lo = p.window_left >= 0 ? max(0, x.rrow[qt] + sk - p.sq - p.window_left) : 0;
                        const int hi = p.causal ? min(L, x.rrow[qt] + sk - p.sq + 1) : L;
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int ka = ka0 + kt * 16 + g4 * 4 + e;
                                if (ka < lo || ka >= hi) s[kt][qt][e] = -INFINITY;
                            }
                    }
                }
                online_block<E, QT, true>(w, s, x.c, x.tb, kb_a + (TILE + krow0) * ROWB, pmax);
            }
        }
    };
    // Interior tiles, software-pipelined over 32-key blocks:  QK^T(b) ; { P.V(b-1)  ||  exp2(b) }.
    // The P.V MFMAs of the previous block and the soft-max VALU work of the current one are independent,
    // so each wave always has matrix and vector work to issue.
    auto run_fast = [&](int t_from, int t_to) {
        if (t_from >= t_to) return;
        if (QT > 2) {
            // 3-tile row blocks (96 accumulator + 48 Q registers) have no room for a second set of
            // probabilities: same schedule without the deferral (their SIMD partner is a 2-tile wave
            // that runs the pipelined form, so matrix and vector phases still interleave on the SIMD)
            for (int t = t_from; t < t_to; ++t) {
                const unsigned kb_a = tile_head(t);
#pragma unroll 1
                for (int blk = 0; blk < p.bpw; ++blk) {
                    const int krow0 = (x.ks * p.bpw + blk) * 32;
                    f32x4 s[2][QT];
                    qk_block<E, QT>(s, qf, x.tb, kb_a + krow0 * ROWB);
                    online_block<E, QT, false>(w, s, x.c, x.tb, kb_a + (TILE + krow0) * ROWB, pmax);
                }
            }
            return;
        }
        typename E::V8 pf_prev[QT];
        int t = t_from, blk = 0;
        unsigned kb_a = tile_head(t);
        unsigned v_prev;
        {   // pipeline fill: first block
            const int krow0 = (x.ks * p.bpw) * 32;
            f32x4 s[2][QT];
            qk_block<E, QT>(s, qf, x.tb, kb_a + krow0 * ROWB);
            softmax_fast<E, QT>(w, s, x.c, pf_prev, pmax);
            v_prev = kb_a + (TILE + krow0) * ROWB;
        }
        while (true) {
            if (++blk == p.bpw) {
                blk = 0;
                if (++t == t_to) break;
                kb_a = tile_head(t);
            }
            const int krow0 = (x.ks * p.bpw + blk) * 32;
            f32x4 s[2][QT];
            qk_block<E, QT>(s, qf, x.tb, kb_a + krow0 * ROWB);
            typename E::V8 pf[QT];
            pv_block<E, QT>(w, pf_prev, x.tb, v_prev);          // MFMA: previous block
            softmax_fast<E, QT>(w, s, x.c, pf, pmax);          // VALU: this block (independent of the P.V above)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) pf_prev[qt] = pf[qt];
            v_prev = kb_a + (TILE + krow0) * ROWB;
        }
        pv_block<E, QT>(w, pf_prev, x.tb, v_prev);      // drain (the slot of tile t_to-1 is still intact)
    };
    for (int attempt = 0; attempt < 2; ++attempt) {
        acc_init<E, QT>(w);
        pmax = 0.f;
        if (x.tid == 0) *redo_flag = 0;
        for (int i = 0; i < LA && t_begin + i < t_end; ++i) dma(t_begin + i);      // LA tiles in flight
        // interior tiles: whole tile inside [lo_max, hi_min); the first tile always primes the max
        int tA = max(t_begin + 1, (lo_max + TILE - 1) / TILE);
        int tB = min(t_end, hi_min / TILE);
        if (attempt == 1 || tB < tA) tA = tB = t_end;
        run_safe(t_begin, tA);
        run_fast(tA, tB);
        run_safe(tB, t_end);
        if (attempt == 1) break;
        // fp16 tops out at 65504: a block sum of 2^14 is two octaves below it
        if (__any(pmax > 16384.f) && x.lane == 0) *redo_flag = 1;
        __syncthreads();
        const int redo = *redo_flag;
        if (redo && x.tid == 0) atomicAdd(&g_attn_redo_count, 1u);
        __syncthreads();
        if (!redo) break;
    }
    // ---- write the (normalised) partial ------------------------------------------
    const int part = split * p.KS + x.ks;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float lt = wave_xor_sum_16_32(w.l[qt]);
        const float inv = lt > 0.f ? 1.f / lt : 0.f;
        const float lse = lt > 0.f ? w.m[qt] * p.scale + __logf(lt) : -INFINITY;
        const int m = x.row0 + qt * 16 + l15;
        if (m < p.M && x.rb < p.RB) {
            const int head = x.kvh * p.g + m / p.sq;
            float* op = p.parts_o + ((((long)part * p.b + x.bi) * p.sq + x.rrow[qt]) * p.H + head) * D + g4 * 4;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4*>(op + dt * 16) = w.acc[dt][qt] * inv;
            if (g4 == 0) p.parts_lse[(((long)part * p.b + x.bi) * p.H + head) * p.sq + x.rrow[qt]] = lse;
        }
    }
}

// ===================== new key block (tree / appended tokens) ===========================
// Kept out of line: a handful of workgroups run it once per launch, and its three
// numerics variants must not weigh on the register allocation of the streaming loop above.
typedef __attribute__((address_space(4))) const AttnK KernArgAttnK;   // the kernel's argument block (constant memory)

template <typename E, int QT, int MODE>
__device__ __attribute__((noinline)) void new_block_path(KernArgAttnK* pk, char* smem) {
    // The parameters are re-read from the kernel-argument segment (scalar loads -> SGPRs): passing the
    // 300-byte block by value would put it on the stack and turn every field access into a scratch load.
    // MODE (= p.new_mode) is a template parameter so that each numerics variant carries only its own state.
    // Round 3: the wave's QT row tiles are processed ONE AFTER THE OTHER (a runtime loop whose body holds the state of a
    // single tile: 16 Q registers, 32 accumulators, 8 mask words) instead of side by side.  Side by side the TARGET variant
    // needed ~480 bytes of scratch per lane (226-230 scratch instructions per instantiation), took ~40 us and was the
    // critical path of a 16k launch; the tiles are independent, and at <= 96 new keys the extra fragment reads are noise.
#if defined(__HIP_DEVICE_COMPILE__)
    const AttnK p = *pk;
#else
    const AttnK p = {};   // host pass of the single-source compile: never executed
    (void)pk;
#endif
    using C = Ctx<E, QT>;
    C x;
    ctx_init(x, p, smem);
    const int L = x.L, l15 = x.l15, g4 = x.g4, tid = x.tid, bi = x.bi, kvh = x.kvh;
    const float c = x.c;
    const int TILE = p.tile;
    const char* kc_base = x.kc_base;
    const char* vc_base = x.vc_base;
    const long kc_row = x.kc_row;
    const int row0 = x.row0;
    const LaneTbl& tb = x.tb;
    const unsigned smem_a = x.smem_a;
    const int n_new = p.n_new;
    const int nblk = (n_new + 31) / 32;
    const char* kn_base = reinterpret_cast<const char*>(p.k_new) + ((long)bi * p.kn_sb + (long)kvh * p.kn_sh) * 2;
    const char* vn_base = reinterpret_cast<const char*>(p.v_new) + ((long)bi * p.kn_sb + (long)kvh * p.kn_sh) * 2;
    const long kn_row = p.kn_ss * 2;
    const int nthreads = blockDim.x;
    const int nwaves = nthreads >> 6;

    // scatter the new K/V rows into the caches (llama.py:396-399, llama_glide.py:312-315)
    if (p.scatter_new && x.chunk == 0) {
        char* kw = reinterpret_cast<char*>(p.k_cache_w) + ((long)bi * p.kc_sb + (long)kvh * p.kc_sh) * 2;
        char* vw = reinterpret_cast<char*>(p.v_cache_w) + ((long)bi * p.kc_sb + (long)kvh * p.kc_sh) * 2;
        const int n_rows = n_new - p.n_new_cached;
        for (int idx = tid; idx < n_rows * 16; idx += nthreads) {
            const int i = idx >> 4, ch = idx & 15;
            const long dst = (long)(L + p.n_new_cached + i) * kc_row;
            reinterpret_cast<uint4*>(kw + dst)[ch] = reinterpret_cast<const uint4*>(kn_base + (long)i * kn_row)[ch];
            reinterpret_cast<uint4*>(vw + dst)[ch] = reinterpret_cast<const uint4*>(vn_base + (long)i * kn_row)[ch];
        }
    }

    // ---- all new keys -> LDS in one go: the ring holds nstages*tile >= 256 keys; K at [0, cap), V behind it
    const int cap = p.nstages * TILE;                 // keys
    const int nkeys = nblk * 32;                      // <= cap (checked on the host)
    char* ldsK = smem;
    char* ldsV = smem + cap * ROWB;
    tile_dma(ldsK, ldsV, x.wave, x.lane, nwaves, nkeys / 4, [&](int key, const char*& kp, const char*& vp) {
        const int j = min(key, n_new - 1);            // tail rows: masked by zero bits
        if (j < p.n_new_cached) {
            kp = kc_base + (long)(L + j) * kc_row;
            vp = vc_base + (long)(L + j) * kc_row;
        } else {
            kp = kn_base + (long)(j - p.n_new_cached) * kn_row;
            vp = vn_base + (long)(j - p.n_new_cached) * kn_row;
        }
    });
    const bool worker = (x.ks == 0) && (x.rb < p.RB);   // the few new keys are not split across key slices
    // keys landed: every wave drains ITS pieces, then the barrier.  The wait has to be spelled out -- the DMA is inline
    // assembly, the compiler does not know that loads are pending, and a wave that only waits for its own mask words
    // behind the barrier reads keys another wave's DMA is still delivering (seen only with thousands of workgroups in
    // flight: a 64-element batch of append calls; one dispatch round was always fast enough)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!worker) return;
    const unsigned vbase0 = smem_a + cap * ROWB;

#pragma unroll 1
    for (int qt = 0; qt < QT; ++qt) {
        const int m = row0 + qt * 16 + l15;                      // this lane's row of the tile (>= M: padding, never stored)
        const int rrow = m < p.M ? m % p.sq : 0;
        const int head = kvh * p.g + (m < p.M ? m / p.sq : 0);
        // Q^T fragments and mask words of the tile (<= 8 words for <= 256 keys)
        typename E::V8 qf[1][4];
        {
            const typename E::T* qp = reinterpret_cast<const typename E::T*>(p.q) + (long)bi * p.q_sb + (long)rrow * p.q_ss +
                                      (long)head * p.q_sh + g4 * 8;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                typename E::V8 v = *reinterpret_cast<const typename E::V8*>(qp + k4 * 32);
                if (MODE == LS_NEW_TARGET && p.prescale_q) {   // `query_states * self.softmax_scale` in the activation dtype (llama.py:407)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = E::from_f32(E::to_f32(v[e]) * p.scale);
                }
                qf[0][k4] = v;
            }
        }
        uint32_t mw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mw[j] = (m < p.M && j < nblk) ? p.mask_bits[((long)bi * p.sq + rrow) * p.mask_words + j] : 0u;
        auto mask_word = [&](int blk) -> uint32_t {
            uint32_t r = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) r = (j == blk) ? mw[j] : r;      // select without dynamic register indexing
            return r;
        };

        WaveAcc<E, 1> w;
        float lse_out, scale_o;
        if (MODE != LS_NEW_TARGET) {
            // ---- blocked online soft-max with base-2 exponentials: flash-attn append semantics, and the Triton tree
            // kernel's loop (BLOCK_N = 32, triton_tree_attn.py:191-235)
            acc_init<E, 1>(w);
            float pmax_unused = 0.f;
#pragma unroll 1
            for (int blk = 0; blk < nblk; ++blk) {
                const uint32_t bits = mask_word(blk);
                f32x4 s[2][1];
                qk_block<E, 1>(s, qf, tb, smem_a + blk * 32 * ROWB);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!((bits >> (kt * 16 + g4 * 4 + e)) & 1u)) s[kt][0][e] = -INFINITY;
                online_block<E, 1, true>(w, s, c, tb, vbase0 + blk * 32 * ROWB, pmax_unused);
            }
            const float lt = wave_xor_sum_16_32(w.l[0]);
            scale_o = lt > 0.f ? 1.0f / lt : 0.f;                                   // acc * (1/l)   (triton_tree_attn.py:242)
            lse_out = lt > 0.f ? w.m[0] * p.scale + logf(lt) : -INFINITY;           // m*scale + ln(l) (:243)
        } else {
            // ---- LlamaAttention.tree_part_fwd numerics (llama.py:406-415): the QK^T result is rounded to the
            // activation dtype, scaled before (last layer, G1) or after the product, soft-max in fp32, probabilities
            // rounded before P.V (G2).  Three sweeps over the (tiny) block: row max, row sum, then P.V.
            float tmax = -INFINITY, tsum = 0.f;
            auto scores = [&](int blk, const f32x4 (&s)[2][1], float (&sv)[8]) {
                const uint32_t bits = mask_word(blk);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float xs = round_to<E>(s[kt][0][e]);
                        if (!p.prescale_q) xs = round_to<E>(xs * p.scale);
                        sv[kt * 4 + e] = ((bits >> (kt * 16 + g4 * 4 + e)) & 1u) ? xs : -INFINITY;
                    }
            };
#pragma unroll 1
            for (int blk = 0; blk < nblk; ++blk) {                     // sweep 0: row max
                f32x4 s[2][1];
                qk_block<E, 1>(s, qf, tb, smem_a + blk * 32 * ROWB);
                float sv[8];
                scores(blk, s, sv);
                float mx = sv[0];
#pragma unroll
                for (int e = 1; e < 8; ++e) mx = fmaxf(mx, sv[e]);
                tmax = fmaxf(tmax, wave_xor_max_16_32(mx));
            }
            const float mref = tmax == -INFINITY ? 0.f : tmax;
#pragma unroll 1
            for (int blk = 0; blk < nblk; ++blk) {                     // sweep 1: row sum of exp(s - max)
                f32x4 s[2][1];
                qk_block<E, 1>(s, qf, tb, smem_a + blk * 32 * ROWB);
                float sv[8];
                scores(blk, s, sv);
#pragma unroll
                for (int e = 0; e < 8; ++e) tsum += expf(sv[e] - mref);
            }
            tsum = wave_xor_sum_16_32(tsum);
            acc_init<E, 1>(w);
            const float den = tsum > 0.f ? tsum : 1.f;
#pragma unroll 1
            for (int blk = 0; blk < nblk; ++blk) {                     // sweep 2: P = dtype(exp(s - max) / sum), P.V
                f32x4 s[2][1];
                qk_block<E, 1>(s, qf, tb, smem_a + blk * 32 * ROWB);
                typename E::V8 pf[1];
                float sv[8];
                scores(blk, s, sv);
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[0][e] = E::from_f32(expf(sv[e] - mref) / den);
                pv_block<E, 1>(w, pf, tb, vbase0 + blk * 32 * ROWB);
            }
            scale_o = 1.f;
            lse_out = tsum > 0.f ? tmax + logf(tsum) : -INFINITY;      // logsumexp (llama.py:415)
        }

        constexpr bool round_o = MODE != LS_NEW_FLASH;     // fp16 matmul result (llama.py:414) / o stored in fp16 (triton :248)
        if (m < p.M) {
            float* op = p.new_o + (((long)bi * p.sq + rrow) * p.H + head) * D + g4 * 4;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                f32x4 o = w.acc[dt][0] * scale_o;
                if (round_o) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = round_to<E>(o[e]);
                }
                *reinterpret_cast<f32x4*>(op + dt * 16) = o;
            }
            if (g4 == 0) p.new_lse[((long)bi * p.H + head) * p.sq + rrow] = lse_out;
        }
    }
}

template <typename E, int QT>
__device__ __forceinline__ void partial_entry(const AttnK& p, char* smem) {
    if (p.has_new && blockIdx.x == 0) {
        KernArgAttnK* pk = (KernArgAttnK*)__builtin_amdgcn_kernarg_segment_ptr();
        if (p.new_mode == LS_NEW_TARGET) new_block_path<E, QT, LS_NEW_TARGET>(pk, smem);
        else if (p.new_mode == LS_NEW_DRAFT) new_block_path<E, QT, LS_NEW_DRAFT>(pk, smem);
        else new_block_path<E, QT, LS_NEW_FLASH>(pk, smem);
    } else {
        prefix_path<E, QT>(p, smem, (int)blockIdx.x - p.has_new);
    }
}

// ===================== warp-specialised prefix path (verification-sized row blocks) =====================
// For 17..20 row tiles (Llama-3 verify: 296 rows) the single-role loop above spends ~1 us per 64 keys in each
// of the three pipes -- MFMA (QK^T + P.V), VALU (exp2 / sums / fp16 packing) and LDS reads -- and runs them
// mostly back to back.  Here the 8 waves split into 4 pairs that share a SIMD: the S wave of a pair computes
// S^T = K.Q^T and the soft-max numerators of its 80 rows and hands P (fp16, in the exact B-operand register
// image) to its O wave through LDS; the O wave accumulates O^T += V^T.P^T.  The O wave's MFMAs run under the S
// wave's VALU work, and inside the S wave the QK^T MFMAs of the NEXT block are issued under the exponentials
// of the current one.  K is read from LDS by 4 waves instead of 8 (V likewise).
//   step j (one 32-key block, one workgroup barrier):   S: QK(j+1) || p = 2^((s(j) - m)c), P(j) -> LDS[j & 1]
//                                                       O: P(j-1) <- LDS[(j-1) & 1], P.V(j-1)
// The reference m of a row is the maximum over the split's first 64 keys (a QK-only look at blocks 0, 1) and
// stays FIXED -- no rescaling of O, so nothing flows back from S to O; lse = m*scale + ln(l) holds for any m.
// If a block sum ever comes within two octaves of the fp16 range the split is redone: a QK-only pass for the
// true row maxima, then the same loop with those.  K and V stream through separate rings of 32-key blocks
// (global_load_lds from all 8 waves, one K and one V piece per wave and step): block b is issued LA steps
// before its K is multiplied (step b-1) and its V slot is released two steps later.
#ifndef LS_WS_HEADROOM
#define LS_WS_HEADROOM 4
#endif
// Diagnostic builds only (tools/build_variant.py abl<N> -DLS_WS_ABLATE=<N>; results are WRONG by construction): parts of the
// warp-specialised kernel's steady loop removed, to price them at the in-round clock (profiles/r5_attn_ceiling.md).
//   1 no K/V DMA behind the first look-ahead blocks     2 no QK^T MFMAs     4 no P.V / row-sum MFMAs
//   8 no soft-max VALU work (a constant P is stored)    16 no LDS fragment reads (K, V^T, P) in the loop
#ifndef LS_WS_ABLATE
#define LS_WS_ABLATE 0
#endif
#ifndef LS_PART_WT
#define LS_PART_WT 1
#endif
// LS_WS_PF = 1 (round 5): the operand fragments of the NEXT step are requested from the LDS in FRONT of the step's barrier and
// are NOT waited for there (lgkmcnt counts in order: the P stores in front of them are) -- the S wave's 8 K fragments of block
// j+2, the O wave's 16 V^T fragment halves of block j -- so that their latency runs under the barrier and both waves have matrix
// work the moment it releases them.  The in-round ablations (profiles/r5_ws_ablations_inround.log) price the exposed fragment
// reads at 34 us of a 165 us call at 128k, as much as either half of the MFMAs.  Costs one block of DMA look-ahead: a block
// must be complete one barrier earlier.  (Round 4's LS_WS_PIPE2 also moved the reads in front of the barrier, but waited for them
// there and moved the DMA issue with them: slower.)
#ifndef LS_WS_PF
#define LS_WS_PF 0
#endif
constexpr int WS_HEADROOM = LS_WS_HEADROOM;     // octaves between the first-64-keys maximum and the fixed soft-max reference
constexpr int WS_QT = 5;                         // row tiles per pair
constexpr int WS_PF = LS_WS_PF;                  // 1: fragments of the next step are fetched in front of the barrier (see LS_WS_PF)
constexpr int WS_LA = 5 + WS_PF;                 // blocks of DMA look-ahead (80 KB of K+V in flight per CU)
// Ring stages.  K: block b lives from step b-1-LA to step b-1;  V: ... to step b+1.  WS_PF reads a block's fragments one barrier
// earlier, so both rings recycle a slot one step earlier and the SAME 14 slots carry one more block of look-ahead (the reads in
// front of a barrier may still be in flight when the first DMA piece behind it is issued: that piece lands >= 1 us later).
constexpr int WS_NK = WS_LA + 1 - WS_PF;
constexpr int WS_NV = WS_LA + 3 - WS_PF;
constexpr int WS_BLK_B = 32 * ROWB;              // 8 KB: 32 keys of K (or V)
constexpr int WS_PBUF_B = 4 * WS_QT * 1024;      // one P buffer: 4 pairs x 5 row tiles x (64 lanes x 16 B)
constexpr int WS_RING_B = (WS_NK + WS_NV) * WS_BLK_B;
constexpr int WS_SAT_WORDS = 64;                 // bitmap of the split's 32-key blocks that held a saturated numerator (<= 2048 blocks)
constexpr int WS_LDS = WS_RING_B + 2 * WS_PBUF_B + 4 * 80 * 4 + 16 + WS_SAT_WORDS * 4 + 4 * 80 * 4;
constexpr int WS_NEW_CAP = WS_RING_B / (2 * ROWB);   // keys the new-block workgroup can hold in the same LDS

// S^T of one 32-key block with ALL 8 K fragments fetched from LDS before the first MFMA (one exposed LDS latency per
// block instead of four); the 40 MFMAs that follow are independent of the caller's soft-max VALU work.
template <typename E, int QT>
__device__ __forceinline__ void qk_block_pf(f32x4 (&s)[2][QT], const typename E::V8 (&qf)[QT][4], const LaneTbl& tb, unsigned kbase) {
    int kx = tb.kx;
    asm volatile("" : "+v"(kx));           // see qk_block
    const unsigned kb = kbase + tb.kb;
    typename E::V8 kf[4][2];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#if LS_WS_ABLATE & 16
            kf[k4][kt] = qf[0][k4];
#else
            kf[k4][kt] = lds_read16<typename E::V8>(kb + ((k4 ^ kx) << 6) + kt * 16 * ROWB);
#endif
        }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#if LS_WS_ABLATE & 2
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) asm volatile("" ::"v"(kf[k4][kt]));       // (the fragment reads stay)
#else
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) s[kt][qt] = E::mfma(kf[k4][kt], qf[qt][k4], s[kt][qt]);
#endif
}

template <typename E, int QT, bool S_ROLE>           // QT row tiles per pair: 5 (17..20 tiles per row block), 3 (two row chunks of 12 tiles)
__device__ __forceinline__ void prefix_path_ws(const AttnK& p, char* smem, int split, int pair) {
    static_assert(QT <= WS_QT, "the P buffers and the reference slots are laid out for WS_QT tiles per pair");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int bi = blockIdx.z, kvh = blockIdx.y % p.Hkv, chunk = blockIdx.y / p.Hkv;
    const int L = p.cache_seqlens[bi];
    const float c = p.scale * LOG2E;
    const int row0 = chunk * p.rows_per_chunk + pair * QT * 16;
    int rrow[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int m = row0 + qt * 16 + l15;
        rrow[qt] = m < p.M ? m % p.sq : 0;        // padding rows: computed, never stored
    }
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned smem_a = (unsigned)(uintptr_t)(lds_char*)smem;
    const LaneTbl tb = make_lane_tbl(l15, g4);
    const char* kc_base = reinterpret_cast<const char*>(p.k_cache) + ((long)bi * p.kc_sb + (long)kvh * p.kc_sh) * 2;
    const char* vc_base = reinterpret_cast<const char*>(p.v_cache) + ((long)bi * p.kc_sb + (long)kvh * p.kc_sh) * 2;
    const long kc_row = p.kc_ss * 2;
    char* pbuf = smem + WS_RING_B;
    float* s_inv = reinterpret_cast<float*>(pbuf + 2 * WS_PBUF_B);
    int* redo_flag = reinterpret_cast<int*>(s_inv + 4 * 80);
    unsigned* sat_bits = reinterpret_cast<unsigned*>(redo_flag + 4);
    float* s_fac = reinterpret_cast<float*>(sat_bits + WS_SAT_WORDS);
    // fp16 only: a soft-max numerator that leaves the fp16 range (a key more than 16 + WS_HEADROOM octaves above the split's
    // reference: a retrieval spike, a sink outside the first 64 keys) converts to 65504 instead of +inf (MODE.FP16_OVFL in the S
    // waves) -- the accumulators stay finite --, the O wave notes its block in `sat_bits` (the row sums jump by >= 65504), and
    // behind the loop the noted blocks are visited once more
    // (`correct_block`): the S wave recomputes their scores, raises the rows' reference to the largest of them and hands the O
    // wave the true numerators minus the 65504 stand-ins, after a rescale of the rows' accumulators.  Round 3 redid the WHOLE
    // split twice (a QK-only pass for the true maxima, then the loop again: 2.7x for that workgroup, and the launch waits for
    // it); now the price is ~3 steps per noted block.  bf16 numerators (exponent range of fp32) cannot saturate.
    // (QT == 5 only: the three-tile instantiation -- fp16 row blocks of 21..24 tiles, none among BASELINE's models, QwQ is bf16 --
    // does not fit the extra state into 256 registers and keeps round 3's redo.)
    constexpr bool SAT_FIX = std::is_same<E, ElemF16>::value && QT == WS_QT;
    if (SAT_FIX && tid < WS_SAT_WORDS) sat_bits[tid] = 0u;      // (visible behind pass_head's barrier)

    // The split's key range in 32-key blocks: ceil(B / n) blocks per split, the last split takes what is left.  (Until round 3
    // the splits were cut at 64-key TILE boundaries, ceil(tiles / n) tiles each: at 16k that is 18 blocks for 28 of the 31 splits,
    // 8 for the next and none for the last two -- 18 steps on the critical path where 17 do.  The even deal
    // [s B / n, (s + 1) B / n) reads better and was measured 3.4x SLOWER: its two divisions change the register allocation of
    // the whole kernel -- 512 spilled registers where this form has none.)
    const int nb_all = (L + 31) / 32;
    const int bps = (nb_all + p.n_splits - 1) / p.n_splits;
    const int b_begin = split * bps;
    const int nblocks = max(0, min(b_begin + bps, nb_all) - b_begin);
    const int last_key = L - 1;

    // one K piece and one V piece per wave: keys 4*wave .. 4*wave+3 of block b (LDS image swizzled on the source side)
#ifdef LS_WS_PROF
    // wall-clock profile of one S wave and one O wave (s_memrealtime: 100 MHz): per-phase sums over the steady steps,
    // dumped into the (unused, prefix-only call) new_o region by workgroup (split 1, kv head 0) -- tools/ws_prof.py
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long pt = 0;
    unsigned long long marks[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // launch timeline of the wave (absolute 100 MHz ticks)
#define WS_MARK(i) do { marks[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    WS_MARK(0);                                                  // entry (after the address set-up above)
#define WS_T0() do { pt = __builtin_amdgcn_s_memrealtime(); } while (0)
#define WS_TS(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); prof[i] += n_ - pt; pt = n_; } while (0)
#else
#define WS_T0()
#define WS_TS(i)
#define WS_MARK(i)
#endif
    // one K piece and one V piece per wave and block.  Inside the cache (all but the split's last blocks) the source address is
    // a wave-uniform base (SGPRs) plus ONE per-lane offset register per operand -- the source-side swizzle of a wave's piece
    // does not depend on the block -- instead of 64-bit per-lane address arithmetic (measured: 106-136 ns of a 1.4 us step)
    const int kq_ = lane >> 4, pos_ = lane & 15;
    const unsigned koff_ = (unsigned)(kq_ * kc_row) + ((pos_ ^ (((wave & 3) << 2) | kq_)) << 4);   // key & 15, key = 4*wave + kq_
    const unsigned voff_ = (unsigned)(kq_ * kc_row) + ((pos_ ^ ((((wave & 1) << 2) | kq_) << 1)) << 4);
    auto dma = [&](int b) {
        const int bg = b_begin + b;
        if (bg * 32 + 32 <= L) {
            const long row = ((long)bg * 32 + wave * 4) * kc_row;
            dma16_s(kc_base + row, koff_, smem_a + (b % WS_NK) * WS_BLK_B + wave * 1024);
            dma16_s(vc_base + row, voff_, smem_a + (WS_NK + b % WS_NV) * WS_BLK_B + wave * 1024);
        } else {
            const int key = wave * 4 + kq_;
            const long ka = min(bg * 32 + key, last_key);                   // tail rows: re-read the last valid key (masked)
            dma16(kc_base + ka * kc_row + ((pos_ ^ (key & 15)) << 4), smem + (b % WS_NK) * WS_BLK_B + wave * 1024);
            dma16(vc_base + ka * kc_row + ((pos_ ^ ((key & 7) << 1)) << 4), smem + (WS_NK + b % WS_NV) * WS_BLK_B + wave * 1024);
        }
    };
    auto k_addr = [&](int b) -> unsigned { return smem_a + (b % WS_NK) * WS_BLK_B; };
    auto v_addr = [&](int b) -> unsigned { return smem_a + (WS_NK + b % WS_NV) * WS_BLK_B; };
    // Both roles execute the same DMA / wait / barrier schedule.  `need` = the youngest block that must have
    // landed when the barrier releases; blocks up to `issued - 1` are in flight.
    auto wait_block = [&](int need, int issued) {
        const int younger = max(0, issued - 1 - need);
        wait_vmcnt(need < nblocks ? 2 * younger : 0);
    };
    auto pass_head = [&]() {
        const int n0 = min(nblocks, WS_NK);        // (WS_PF: block LA follows behind the look's barrier, `look_done`)
        for (int b = 0; b < n0; ++b) dma(b);
        wait_block(1 + WS_PF, n0);                 // blocks 0 and 1 (reference look, first QK) [+ block 2: read in front of barrier 0]
        __builtin_amdgcn_s_barrier();
    };
    auto step_head = [&](int j) {
#if !(LS_WS_ABLATE & 1)
        if (j + 1 + WS_LA < nblocks) dma(j + 1 + WS_LA);                // its K slot died at step j-1, its V slot at step j-1
#endif
    };
    auto step_tail = [&](int j) {
        wait_block(j + 2 + WS_PF, min(nblocks, j + 2 + WS_LA));         // K(j+2) is multiplied at step j+1 (WS_PF: K(j+3) is read in front of barrier j+1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // P writes / reads of this step are done
        __builtin_amdgcn_s_barrier();
    };
    const unsigned p_base = smem_a + WS_RING_B + pair * QT * 1024 + lane * 16;

    if constexpr (S_ROLE) {
        // MODE.FP16_OVFL = 1 in the S waves: a numerator beyond the fp16 range converts to 65504 instead of +inf (true
        // infinities stay), at no instruction -- the saturation "clamp" of the scheme described at `sat_bits`
        if constexpr (SAT_FIX) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
        typename E::V8 qf[QT][4];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int m = row0 + qt * 16 + l15;
            const int head = kvh * p.g + (m < p.M ? m / p.sq : 0);
            const typename E::T* qp = reinterpret_cast<const typename E::T*>(p.q) + (long)bi * p.q_sb + (long)rrow[qt] * p.q_ss +
                                      (long)head * p.q_sh + g4 * 8;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                qf[qt][k4] = *reinterpret_cast<const typename E::V8*>(qp + k4 * 32);
                // Padding rows (rows m >= M of the last tile: 24 of the 320 a Llama-3 verify pass multiplies) carry ZERO queries
                // and, below, an infinite reference -- their scores and numerators are exact zeros.  The kernel runs at the clock
                // the power budget allows (the same launch takes 167 us on all-zero operands and 232 us on random ones,
                // profiles/r3_power_bound.json), and an MFMA on zeros costs next to nothing.
                if (m >= p.M)
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[qt][k4][e] = E::from_f32(0.f);
            }
        }
        float mref[QT];
        auto mask_tail = [&](f32x4 (&s)[2][QT], int b) {    // keys >= L of a block that crosses the end of the cache
            const int ka0 = (b_begin + b) * 32;
            if (ka0 + 32 > L) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ka0 + kt * 16 + g4 * 4 + e >= L) s[kt][qt][e] = -INFINITY;
            }
        };
        auto row_max = [&](const f32x4 (&s)[2][QT], float (&mx)[QT]) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const float v = fmaxf(fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), fmaxf(s[0][qt][2], s[0][qt][3])),
                                      fmaxf(fmaxf(s[1][qt][0], s[1][qt][1]), fmaxf(s[1][qt][2], s[1][qt][3])));
                mx[qt] = fmaxf(mx[qt], wave_xor_max_16_32(v));
            }
        };
        // mode 0: reference from the first 64 keys, fixed;  1: QK-only pass for the true row maxima;  2: as 0 with mref given
        auto run_pass = [&](int mode) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
                if (mode != 2) mref[qt] = -INFINITY;
            pass_head();
            WS_MARK(1);                            // blocks 0 and 1 have landed (Q loads issued before them)
            f32x4 s_cur[2][QT];
            if (mode == 0 && nblocks > 1) {        // block 1's share of the reference (block 0's follows)
                qk_block<E, QT>(s_cur, qf, tb, k_addr(1));
                mask_tail(s_cur, 1);
                row_max(s_cur, mref);
            }
            if (nblocks > 0) {
                qk_block<E, QT>(s_cur, qf, tb, k_addr(0));
                mask_tail(s_cur, 0);
                if (mode == 0) row_max(s_cur, mref);
            }
            if (mode == 0) {
                // Head-room (round 4): the fixed reference sits WS_HEADROOM octaves ABOVE the maximum of the first 64 keys, so a
                // later key may exceed that maximum by (16 + WS_HEADROOM) octaves before its numerator leaves the fp16 range
                // and the split is redone.  All numerators, their fp16 roundings and the row sums scale by exactly 2^-WS_HEADROOM
                // (only values within WS_HEADROOM octaves of the smallest normal number -- weights below 2^-(14 - WS_HEADROOM)
                // of the first-64 maximum -- lose bits earlier), o = acc / l and lse = m * scale + ln(l) do not move.  Measured
                // (profiles/r4_redo_*.json): 8 retrieval-style keys 14 nats above the bulk redo 64 workgroups of a 128k call
                // (427 us instead of 226) at WS_HEADROOM 0 and none at 4.
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mref[qt] += (float)WS_HEADROOM / c;        // (-inf stays -inf)
            }
            if (mode != 1) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    if (row0 + qt * 16 + l15 >= p.M) mref[qt] = INFINITY;      // padding rows: p = 2^(0 - inf) = 0
            }
            typename E::V8 kfr[4][2];              // WS_PF: the K fragments of the block whose scores the next step forms
            auto load_kf = [&](unsigned kbase) {
                int kx = tb.kx;
                asm volatile("" : "+v"(kx));       // see qk_block
                const unsigned kb = kbase + tb.kb;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) kfr[k4][kt] = lds_read16<typename E::V8>(kb + ((k4 ^ kx) << 6) + kt * 16 * ROWB);
            };
            auto qk_from_kfr = [&](f32x4 (&sx)[2][QT]) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) sx[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) sx[kt][qt] = E::mfma(kfr[k4][kt], qf[qt][k4], sx[kt][qt]);
            };
            if (WS_PF && nblocks > 1) load_kf(k_addr(1));                     // (block 1 is complete since pass_head's barrier)
            __builtin_amdgcn_s_barrier();          // K(0) is consumed: step 0 may overwrite its slot
            if (WS_PF && WS_LA < nblocks) dma(WS_LA);      // (the ring's last look-ahead block takes K(0)'s slot)
            WS_MARK(2);                            // reference look done
            typedef __attribute__((address_space(3))) typename E::V8 lds_v8;
            // soft-max numerators of block j (reference mref, fixed) -> P(j) in LDS.  The row sums are NOT formed here: the
            // O wave gets them from the matrix pipe (a ninth V^T tile whose row 0 is all ones: l = sum of the fp16 P it
            // multiplies -- the normaliser of exactly the numerators used), which takes 40 adds and the overflow watch off
            // this wave's critical VALU path; a P that overflowed fp16 shows up there as a non-finite l and redoes the split
            auto softmax_store = [&](int j, const f32x4 (&sc)[2][QT]) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const float mc = mref[qt] * c;
                    typename E::V8 pf;
#if LS_WS_ABLATE & 8
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = E::from_f32(mc * 0.f + 0.001f * (float)(e + 1));
                    asm volatile("" ::"v"(sc[0][qt]), "v"(sc[1][qt]));
#else
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            pf[kt * 4 + e] = E::from_f32(__builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][qt][e], c, -mc)));
#endif
#ifdef LS_MUTATE_SKIP_BLOCK
                    // MUTANT (tests/test_gpu_ops.py::test_mutant_is_caught, never the product build): one 32-key block of one
                    // split of one kv head contributes nothing
                    if (j == 5 && split == 3 && kvh == 1)
#pragma unroll
                        for (int e = 0; e < 8; ++e) pf[e] = E::from_f32(0.f);
#endif
                    *(lds_v8*)(uintptr_t)(p_base + (j & 1) * WS_PBUF_B + qt * 1024) = pf;
                }
            };
            if (mode == 1) {                               // QK-only pass: true row maxima
#pragma unroll 1
                for (int j = 0; j <= nblocks; ++j) {
                    step_head(j);
                    if (j < nblocks) {
                        row_max(s_cur, mref);
                        if (j + 1 < nblocks) {
                            if constexpr (WS_PF) {         // (the rings recycle K(j+1)'s slot at this step's head: fragments in registers)
                                qk_from_kfr(s_cur);
                                if (j + 2 < nblocks) load_kf(k_addr(j + 2));
                            } else {
                                qk_block_pf<E, QT>(s_cur, qf, tb, k_addr(j + 1));
                            }
                            mask_tail(s_cur, j + 1);
                        }
                    }
                    step_tail(j);
                }
            } else {
                // steady state: ONE basic block holds the QK^T MFMAs of block j+1 and the VALU work of block j, so that
                // the scheduler can lay them out as asked below: an MFMA, then the VALU instructions its 16 cycles hide
                auto step_ab = [&](int j, f32x4 (&s_in)[2][QT], f32x4 (&s_next)[2][QT], auto masked, auto fixed_wait) {
                    WS_T0();
                    step_head(j);
                    WS_TS(0);
                    if constexpr (WS_PF) {
                        // S^T of block j+1 from the fragments fetched in front of the last barrier, under the exponentials of block j
                        qk_from_kfr(s_next);
                        softmax_store(j, s_in);
#pragma unroll
                        for (int i = 0; i < 8 * QT; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);     // 6 VALU
                        }
                        if constexpr (decltype(masked)::value) mask_tail(s_next, j + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (j + 2 < nblocks) load_kf(k_addr(j + 2));               // complete since barrier j-1; waited for by its first MFMA
                    } else {
                        qk_block_pf<E, QT>(s_next, qf, tb, k_addr(j + 1));
                        softmax_store(j, s_in);
                        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);         // all 8 K-fragment LDS reads first,
                        __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);        // VALU work while they are in flight
#pragma unroll
                        for (int i = 0; i < 8 * QT; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);     // 6 VALU
                        }
                        if constexpr (decltype(masked)::value) mask_tail(s_next, j + 1);
                    }
                    WS_TS(1);
                    // K(j+2) is multiplied at step j+1.  In the steady state exactly LA-1 younger blocks (2 pieces each) are in
                    // flight: one immediate wait instead of the compare / branch ladder of wait_block (measured: 155-170 ns of
                    // a 1.4 us step went into that ladder)
                    if constexpr (decltype(fixed_wait)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WS_LA - 1 - WS_PF)) : "memory");
                    else wait_block(j + 2 + WS_PF, min(nblocks, j + 2 + WS_LA));
                    WS_TS(2);
                    // the P stores are done (LDS operations of a wave complete in order: the 8 fragment reads behind them may still fly)
                    if (WS_PF && j + 2 < nblocks) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    WS_TS(3);
                    __builtin_amdgcn_s_barrier();
                    WS_TS(4);
                };
                // only the split's LAST block can cross the end of the cache: its masking (35 selects per lane when the
                // compiler if-converts it into every iteration) is peeled off the loop
                // one step with the scores copied back (edges of the loop), and the steady loop two steps at a time with the two
                // score register sets swapping roles -- no 40-register copy per step
                auto steady_step = [&](int j, auto masked, auto fixed_wait) {
                    f32x4 s_next[2][QT];
                    step_ab(j, s_cur, s_next, masked, fixed_wait);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) s_cur[kt][qt] = s_next[kt][qt];
                };
                const int n_fixed = max(0, nblocks - 1 - WS_LA);      // steps j with j + 2 + LA <= nblocks
                int j0 = 0;
                {
                    f32x4 s_alt[2][QT];
#pragma unroll 1
                    for (; j0 + 2 <= n_fixed; j0 += 2) {
                        step_ab(j0, s_cur, s_alt, std::false_type{}, std::true_type{});
                        step_ab(j0 + 1, s_alt, s_cur, std::false_type{}, std::true_type{});
                    }
                }
#pragma unroll 1
                for (int j = j0; j < n_fixed; ++j) steady_step(j, std::false_type{}, std::true_type{});
#pragma unroll 1
                for (int j = n_fixed; j + 2 < nblocks; ++j) steady_step(j, std::false_type{}, std::false_type{});
                if (nblocks > 1) steady_step(nblocks - 2, std::true_type{}, std::false_type{});
                if (nblocks > 0) {                         // last block: nothing left to multiply
                    step_head(nblocks - 1);
                    softmax_store(nblocks - 1, s_cur);
                    step_tail(nblocks - 1);
                }
                step_head(nblocks);                        // the O waves' last P.V
                step_tail(nblocks);
            }
        };
        run_pass(0);
        WS_MARK(3);                                        // step loop done
        if (tid == 0) *redo_flag = 0;
        __syncthreads();
        __syncthreads();                                   // (the O waves raise the flag in between)
        if (*redo_flag) {
            if (tid == 0) atomicAdd(&g_attn_redo_count, 1u);
            // (a split redone with its true row maxima -- a numerator beyond fp32, > 88 nats above the reference -- needs no
            // correction: the notes of pass 0 are void.  A flag carried across the two passes cost 688 bytes of scratch.)
            if (SAT_FIX && tid < WS_SAT_WORDS) sat_bits[tid] = 0u;
            __syncthreads();
            run_pass(1);
            run_pass(2);
        }
#ifdef LS_WS_PROF
        WS_MARK(4);
        if (!p.has_new && split == 1 && kvh == 0 && bi == 0 && wave == 0 && lane < 8)
            reinterpret_cast<unsigned long long*>(p.new_o)[16 + lane] = marks[0] * (lane == 0) + marks[1] * (lane == 1) + marks[2] * (lane == 2) +
                                                                        marks[3] * (lane == 3) + marks[4] * (lane == 4);
        if (!p.has_new && split == 1 && kvh == 0 && bi == 0 && wave == 0 && lane < 6)
            reinterpret_cast<unsigned long long*>(p.new_o)[lane] = prof[0] * (lane == 0) + prof[1] * (lane == 1) + prof[2] * (lane == 2) +
                                                                   prof[3] * (lane == 3) + prof[4] * (lane == 4) + (unsigned long long)nblocks * (lane == 5);
#endif
        float mcur[QT];                                    // the rows' reference after the corrections below (= mref without)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mcur[qt] = mref[qt];
        if constexpr (SAT_FIX) {
            typedef __attribute__((address_space(3))) typename E::V8 lds_v8c;
            const int nwords = min(WS_SAT_WORDS, (nblocks + 31) >> 5);
            for (int w = 0; w < nwords; ++w) {
                unsigned bits = __builtin_amdgcn_readfirstlane(sat_bits[w]);
                while (bits) {
                    const int b = w * 32 + __builtin_ctz(bits);
                    bits &= bits - 1;
                    // ---- correct_block(b), S role: barriers B1 (block landed), B2 (P and the row factors are in LDS), B3 (consumed)
                    dma(b);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    f32x4 sc[2][QT];
                    qk_block<E, QT>(sc, qf, tb, k_addr(b));
                    mask_tail(sc, b);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const float m0c = mref[qt] * c;
                        bool sat[8];
                        float smax = -INFINITY;
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                // exactly the loop's numerator: it was clamped <=> its fp32 value rounds beyond 65504
                                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][qt][e], c, -m0c));
                                sat[kt * 4 + e] = p0 >= 65520.0f;
                                if (sat[kt * 4 + e]) smax = fmaxf(smax, sc[kt][qt][e]);
                            }
                        smax = wave_xor_max_16_32(smax);
                        const float m_new = fmaxf(mcur[qt], smax);
                        const float fac = m_new == mcur[qt] ? 1.0f : __builtin_amdgcn_exp2f((mcur[qt] - m_new) * c);
                        const float stand = 65504.0f * __builtin_amdgcn_exp2f((mref[qt] - m_new) * c);     // what the loop added per saturated key, in the new scale
                        typename E::V8 pf;
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                pf[kt * 4 + e] = E::from_f32(sat[kt * 4 + e] ? __builtin_amdgcn_exp2f((sc[kt][qt][e] - m_new) * c) - stand : 0.0f);
                        *(lds_v8c*)(uintptr_t)(p_base + qt * 1024) = pf;
                        if (g4 == 0) s_fac[pair * 80 + qt * 16 + l15] = fac;
                        mcur[qt] = m_new;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
        // the reference of every row, for the O wave's log-normaliser lse = m*scale + ln(l)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
            if (g4 == 0) s_inv[pair * 80 + qt * 16 + l15] = mcur[qt] * p.scale;
        __syncthreads();
    } else {
        f32x4 acc[8][QT];
        f32x4 lacc[QT];                            // row 0 of the ones tile: l of query row l15 in the lanes with g4 == 0, element 0
        typename E::V8 ones;                       // A operand of that tile: row 0 (lanes with l15 == 0) = 1 for every key
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = E::from_f32(l15 == 0 ? 1.f : 0.f);
        float sat_prev = 0.f;                      // the five rows' total of the previous step (saturation watch)
        // The bitmap covers WS_SAT_WORDS * 32 = 2048 blocks (65536 keys) of a split.  A saturated numerator in a LATER block (a
        // forced n_splits = 1 or a batched prompt call over > 64k keys per split) cannot be noted -- and since FP16_OVFL clamps
        // instead of producing inf, nothing else would catch it: it takes the true-maxima redo instead.
        bool sat_far = false;                      // wave-uniform
        auto run_pass = [&](int mode) {
            sat_prev = 0.f;
            sat_far = false;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                lacc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            pass_head();
            __builtin_amdgcn_s_barrier();          // the S waves' look at blocks 0 and 1 is over
            if (WS_PF && WS_LA < nblocks) dma(WS_LA);
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            union VF {
                struct { s16x4 a, b; } s;
                typename E::V8 v;
            } vf[8];                               // all 8 V^T fragments before the first MFMA (WS_PF: fetched in front of the last barrier)
            auto load_vf = [&](int blk) {
                int vx = tb.vx;
                asm volatile("" : "+v"(vx));       // see qk_block
                const unsigned vb = v_addr(blk) + tb.vb;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    const unsigned va = vb + ((dt ^ vx) << 5);
#if LS_WS_ABLATE & 16
                    vf[dt].v = ones;
                    asm volatile("" ::"v"(va));
#else
                    vf[dt].s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)va);
                    vf[dt].s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(va + 16 * ROWB));
#endif
                }
            };
#pragma unroll 1
            for (int j = 0; j <= nblocks; ++j) {
                WS_T0();
                step_head(j);
                WS_TS(0);
                if (mode != 1 && j >= 1) {
                    const int jj = j - 1;
                    typename E::V8 pf[QT];
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
#if LS_WS_ABLATE & 16
                        pf[qt] = ones;
#else
                        pf[qt] = lds_read16<typename E::V8>(p_base + (jj & 1) * WS_PBUF_B + qt * 1024);
#endif
                    }
                    if constexpr (!WS_PF) load_vf(jj);
#if LS_WS_ABLATE & 4
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) asm volatile("" ::"v"(vf[dt].v));
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) asm volatile("" ::"v"(pf[qt]));
#else
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) acc[dt][qt] = E::mfma(vf[dt].v, pf[qt], acc[dt][qt]);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) lacc[qt] = E::mfma(ones, pf[qt], lacc[qt]);
#endif
                    if constexpr (SAT_FIX) {
                        // a saturated numerator adds >= 65504 to its row's sum: one comparison of the five rows' running total
                        // per step (7 vector instructions; any VALU work in this wave comes out of the SIMD the pair shares --
                        // clamping and scanning the numerators here, 48 instructions per step, cost 12 % of the kernel).
                        // A step whose ordinary numerators add up to as much is noted too: its correction finds nothing to do.
                        float tot = lacc[0][0];
#pragma unroll
                        for (int qt = 1; qt < QT; ++qt) tot += lacc[qt][0];
                        if (__any(tot - sat_prev >= 65504.0f)) {
                            if (jj >= WS_SAT_WORDS * 32) sat_far = true;
                            else if (lane == 0)
                                __hip_atomic_fetch_or(sat_bits + (jj >> 5), 1u << (jj & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        sat_prev = tot;
                    }
                }
                if constexpr (WS_PF) {
                    // V^T of block j, multiplied at step j+1 (the block is complete since barrier j-2): requested here, behind
                    // this step's MFMAs, and not waited for in front of the barrier
                    __builtin_amdgcn_sched_barrier(0);
                    if (mode != 1 && j < nblocks) load_vf(j);
                }
                WS_TS(1);
                // steady state: exactly LA-1 younger blocks (2 pieces each) are in flight behind K(j+2) -- see the S role
                if (j + 2 + WS_LA <= nblocks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WS_LA - 1 - WS_PF)) : "memory");
                else wait_block(j + 2 + WS_PF, min(nblocks, j + 2 + WS_LA));
                WS_TS(2);
                // (WS_PF: the P fragments of this step were consumed by its MFMAs; only the fragment reads above are outstanding)
                if constexpr (!WS_PF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                WS_TS(3);
                __builtin_amdgcn_s_barrier();
                WS_TS(4);
            }
        };
        run_pass(0);
        WS_MARK(3);
        __syncthreads();
        {   // an fp16 P that overflowed (or an inf - inf behind it) leaves a non-finite row sum: redo with the true row maxima
            bool bad = false;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) bad |= !(fabsf(lacc[qt][0]) <= 3.0e38f);
            if ((__any(bad && g4 == 0) || sat_far) && lane == 0) *redo_flag = 1;
        }
        __syncthreads();
        if (*redo_flag) {
            __syncthreads();
            run_pass(1);
            run_pass(2);
        }
#ifdef LS_WS_PROF
        if (!p.has_new && split == 1 && kvh == 0 && bi == 0 && wave == 4 && lane < 6)
            reinterpret_cast<unsigned long long*>(p.new_o)[8 + lane] = prof[0] * (lane == 0) + prof[1] * (lane == 1) + prof[2] * (lane == 2) +
                                                                       prof[3] * (lane == 3) + prof[4] * (lane == 4) + (unsigned long long)nblocks * (lane == 5);
#endif
        if constexpr (SAT_FIX) {
            const int nwords = min(WS_SAT_WORDS, (nblocks + 31) >> 5);
            for (int w = 0; w < nwords; ++w) {
                unsigned bits = __builtin_amdgcn_readfirstlane(sat_bits[w]);
                while (bits) {
                    const int b = w * 32 + __builtin_ctz(bits);
                    bits &= bits - 1;
                    // ---- correct_block(b), O role
                    dma(b);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_barrier();          // B2: the S wave's numerators and row factors
                    typename E::V8 pf[QT];
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const float fac = s_fac[pair * 80 + qt * 16 + l15];
                        pf[qt] = lds_read16<typename E::V8>(p_base + qt * 1024);
                        lacc[qt] *= fac;
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt) acc[dt][qt] *= fac;
                    }
                    typedef __attribute__((address_space(3))) s16x4 lds_s16x4c;
                    int vx = tb.vx;
                    asm volatile("" : "+v"(vx));
                    const unsigned vb = v_addr(b) + tb.vb;
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) {
                        union {
                            struct { s16x4 a, b; } s;
                            typename E::V8 v;
                        } vf;
                        const unsigned va = vb + ((dt ^ vx) << 5);
                        vf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)(uintptr_t)va);
                        vf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)(uintptr_t)(va + 16 * ROWB));
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) acc[dt][qt] = E::mfma(vf.v, pf[qt], acc[dt][qt]);
                    }
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) lacc[qt] = E::mfma(ones, pf[qt], lacc[qt]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();          // B3
                }
            }
        }
        __syncthreads();                           // the pair's m*scale is in LDS
        WS_MARK(5);                                // ready to write the partial
#if LS_PART_WT
        // this (split, batch element)'s slab of the partials as a buffer resource: [sq][H][D] fp32, < 4 GB by construction
        const __amdgpu_buffer_rsrc_t part_rs = __builtin_amdgcn_make_buffer_rsrc(
            p.parts_o + ((long)split * p.b + bi) * p.sq * p.H * D, 0, p.sq * p.H * D * 4, 0x00020000);
#endif
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const float lt = __shfl(lacc[qt][0], l15);       // (lanes g4 == 0 hold it)
            const float inv = lt > 0.f ? 1.f / lt : 0.f;
            const int m = row0 + qt * 16 + l15;
            if (m < p.M) {
                const int head = kvh * p.g + m / p.sq;
                if (g4 == 0)
                    p.parts_lse[(((long)split * p.b + bi) * p.H + head) * p.sq + rrow[qt]] =
                        lt > 0.f ? s_inv[pair * 80 + qt * 16 + l15] + __logf(lt) : -INFINITY;
                float* op = p.parts_o + ((((long)split * p.b + bi) * p.sq + rrow[qt]) * p.H + head) * D + g4 * 4;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
#if LS_PART_WT
                    // the partials leave the L2 as they are written (agent-scope write-through, `sc1` = aux 16) instead of in one
                    // write-back burst at the end of the kernel: 37.6 MB of dirty lines cost the launch ~2 us at its end
                    // (round 5 A/B inside the round, profiles/r5_part_wt.json; -DLS_PART_WT=0 rebuilds the plain-store variant;
                    // `nt` stores were slower, round 4).  Through the compiler's buffer-store builtin, NOT inline asm: a VALU
                    // write of a > 64-bit store's data registers needs two wait states behind the store, which the compiler only
                    // inserts for stores it knows -- the asm form of this line gave wrong partials in one build and right ones
                    // in another.
                    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
                    const f32x4 v_ = acc[dt][qt] * inv;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v_), part_rs,
                                                           (unsigned)((((long)rrow[qt] * p.H + head) * D + g4 * 4 + dt * 16) * 4), 0, 16);
#else
                    *reinterpret_cast<f32x4*>(op + dt * 16) = acc[dt][qt] * inv;
#endif
                }
            }
        }
#ifdef LS_WS_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WS_MARK(6);                                // partial stores acknowledged
        if (!p.has_new && split == 1 && kvh == 0 && bi == 0 && wave == 4 && lane < 8)
            reinterpret_cast<unsigned long long*>(p.new_o)[24 + lane] = marks[0] * (lane == 0) + marks[3] * (lane == 3) +
                                                                        marks[5] * (lane == 5) + marks[6] * (lane == 6);
#endif
    }
}

template <typename E, int QT>
__global__ __launch_bounds__(MAX_THREADS) void attn_partial_ws_kernel(const AttnK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.has_new && blockIdx.x == 0) {            // the new-key block keeps the 3,3,3,3,2,2,2,2 row split
        KernArgAttnK* pk = (KernArgAttnK*)__builtin_amdgcn_kernarg_segment_ptr();
        const int rb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (rb < p.rbA) {
            if (p.new_mode == LS_NEW_TARGET) new_block_path<E, 3, LS_NEW_TARGET>(pk, smem);
            else if (p.new_mode == LS_NEW_DRAFT) new_block_path<E, 3, LS_NEW_DRAFT>(pk, smem);
            else new_block_path<E, 3, LS_NEW_FLASH>(pk, smem);
        } else {
            if (p.new_mode == LS_NEW_TARGET) new_block_path<E, 2, LS_NEW_TARGET>(pk, smem);
            else if (p.new_mode == LS_NEW_DRAFT) new_block_path<E, 2, LS_NEW_DRAFT>(pk, smem);
            else new_block_path<E, 2, LS_NEW_FLASH>(pk, smem);
        }
    } else {
        // Roles by wave index: S = waves 0-3, O = waves 4-7, pair = wave & 3.  The dispatcher places the 8 waves of a
        // workgroup round-robin over the 4 SIMDs, so the S and the O wave of a pair share one (the O wave's MFMAs run under
        // the S wave's VALU work).  Drawing the roles from the SIMD id actually read (HW_ID) costs three barriers and
        // measured 1.8 us SLOWER per launch at 16k, 6 us at 128k.
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (w < 4) prefix_path_ws<E, QT, true>(p, smem, (int)blockIdx.x - p.has_new, w);
        else prefix_path_ws<E, QT, false>(p, smem, (int)blockIdx.x - p.has_new, w - 4);
    }
    drain_lds_dma();
}

// The two measured negatives of round 3 (four-wave and ping-pong kernels) live in tools/mb/ and are only compiled into
// diagnostic variants (tools/build_variant.py w4 -DLS_WITH_W4 / pp -DLS_WITH_PP); the product never dispatches them.
#ifdef LS_WITH_W4
#include "../../tools/mb/attn_w4_kernel.inc"
#endif
#ifdef LS_WITH_PP
#include "../../tools/mb/attn_pp_kernel.inc"
#endif

// Row blocks 0..rbA-1 carry QTA tiles, the rest QTB: both instantiations execute the same barrier
// sequence (identical tile loop), so a workgroup may mix them wave by wave.
template <typename E, int QTA, int QTB>
__global__ __launch_bounds__(MAX_THREADS) void attn_partial_kernel(const AttnK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (QTA == QTB) {
        partial_entry<E, QTA>(p, smem);
    } else {
        const int rb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) / p.KS;
        if (rb < p.rbA) partial_entry<E, QTA>(p, smem);
        else partial_entry<E, QTB>(p, smem);
    }
    drain_lds_dma();
}

// ---- stage 2: combine + merge ----------------------------------------------------------
struct FinK {
    const float* parts_o;    // [n_parts][b][sq][H][D]
    const float* parts_lse;  // [n_parts][b][H][sq]
    const float* new_o;      // [b][sq][H][D] or null
    const float* new_lse;
    void* out;               // dtype [b,sq,H,D] (strided) or null
    float* o32;              // fp32 [b][sq][H][D] or null
    float* lse_out;          // [b][H][sq] or null
    int n_parts, b, sq, H, mode;   // mode = LS_NEW_*
    long part_o_stride, part_lse_stride;   // elements between consecutive parts
    long out_sb, out_ss, out_sh;
    // sequence-sharded prefix (xgmi.hip): x_mode 1 = the (o32, lse) record goes into slot [parity][rank] of every peer's
    // mailbox, a flag follows it and the epoch moves on; 2 = the parts ARE the mailbox slots [parity][0..world) of this
    // rank, read once every peer's flag of the epoch just pushed is up
    int x_mode, x_rank, x_world;
    long x_cap, x_lse_off;                 // floats per slot; offset of the lse block inside a record
    char* const* x_peers;
    const char* x_box;
    XCtl* x_ctl;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename E, int XM>                      // XM = FinK::x_mode, a compile-time copy (0: no exchange code at all)
__global__ __launch_bounds__(256) void attn_finish_kernel(const FinK p) {
    const int tid = threadIdx.x;
    const int r_ = blockIdx.x * 8 + (tid >> 5);
    const bool act = r_ < p.sq;                    // rows of the last block beyond sq: no stores
    if (!act && XM == 0) return;
    const int r = act ? r_ : p.sq - 1;             // (exchange modes: every thread stays for the barriers)
    const int h = blockIdx.y, bi = blockIdx.z;
    const int d = (tid & 31) * 4;
    const float* parts_o_ = p.parts_o;
    const float* parts_lse_ = p.parts_lse;
    unsigned long long epoch = 0;
    int parity = 0;
    if constexpr (XM != 0) {                       // (no thread has left yet)
        epoch = p.x_ctl->epoch - (XM == 2 ? 1 : 0);        // (the push in front of a wait has moved it on)
        parity = (int)(epoch & 1);
        if constexpr (XM == 2) {
            if (tid < p.x_world) xchg_wait_flag(p.x_box, p.x_ctl, parity, tid, epoch);
            __syncthreads();
            parts_o_ = reinterpret_cast<const float*>(p.x_box + XCHG_DATA_OFF) + (long)parity * p.x_world * p.x_cap;
            parts_lse_ = parts_o_ + p.x_lse_off;
            if (!act) return;                      // (behind the barrier)
        }
    }
    const long lse_idx = ((long)bi * p.H + h) * p.sq + r;
    const long o_idx = (((long)bi * p.sq + r) * p.H + h) * D + d;
    const long part_lse_stride = p.part_lse_stride;
    const long part_o_stride = p.part_o_stride;
    const bool joint = (p.mode == LS_NEW_FLASH) && p.new_o != nullptr;
    // XM == 2: the parts are mailbox slots that PEERS wrote over xGMI.  The mailbox is uncached device memory, but nothing in the
    // memory model keeps a line of it out of this CU's L1 (a stale line of epoch e - 2 of the same parity slot would be merged
    // silently), and `nt` is only a streaming HINT on gfx94x/gfx950 -- L1 bypass is a property of the scope bits.  The
    // records are therefore read with SYSTEM-scope loads (`sc0 sc1`, aux = 17 of the raw buffer load: served beyond the L1 and
    // the non-coherent L2 lines), through the compiler's builtin so that it still counts them.  The relaxed flag poll in front
    // (xchg_wait_flag) orders nothing by itself: the loads are issued behind it and behind the caller's barrier.
    __amdgpu_buffer_rsrc_t box_rs;
    if constexpr (XM == 2) box_rs = xchg_record_rsrc(parts_o_, (size_t)p.x_world * p.x_cap * 4);
    auto ld4 = [&](const float* a_) -> f32x4 {
        if constexpr (XM == 2) return xchg_load16(box_rs, (unsigned)((const char*)a_ - (const char*)parts_o_));
        else return *reinterpret_cast<const f32x4*>(a_);
    };
    auto ld1 = [&](const float* a_) -> float {
        if constexpr (XM == 2) return xchg_load4(box_rs, (unsigned)((const char*)a_ - (const char*)parts_o_));
        else return *a_;
    };

    float lnew = -INFINITY;
    if (p.new_o != nullptr) lnew = p.new_lse[lse_idx];
    float mref, den = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    const int n = p.n_parts;
    if (n >= 1 && n <= 32) {
        // every load of the merge is issued before the first use: the kernel is one wave per SIMD and pure latency,
        // so the number of dependent load batches is its run time.  Parts beyond n re-read part n-1 with weight 0.
        float l[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) l[u] = ld1(parts_lse_ + (u < n ? u : n - 1) * part_lse_stride + lse_idx);
        f32x4 oa[16], ob[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            oa[u] = ld4(parts_o_ + (u < n ? u : n - 1) * part_o_stride + o_idx);
        if (n > 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                ob[u] = ld4(parts_o_ + (16 + u < n ? 16 + u : n - 1) * part_o_stride + o_idx);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 32; ++u) mx = fmaxf(mx, u < n ? l[u] : -INFINITY);
        if (joint) mx = fmaxf(mx, lnew);
        mref = mx == -INFINITY ? 0.f : mx;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float wgt = u < n ? expf(l[u] - mref) : 0.f;
            den += wgt;
            o += oa[u] * wgt;
        }
        if (n > 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float wgt = 16 + u < n ? expf(l[16 + u] - mref) : 0.f;
                den += wgt;
                o += ob[u] * wgt;
            }
        }
    } else {
        // pass 1: reference max over the parts (independent scalar loads)
        float mx = -INFINITY;
#pragma unroll 8
        for (int i = 0; i < p.n_parts; ++i) mx = fmaxf(mx, ld1(parts_lse_ + i * part_lse_stride + lse_idx));
        if (joint) mx = fmaxf(mx, lnew);
        mref = mx == -INFINITY ? 0.f : mx;
        // pass 2: weighted sum in fixed part order; branch-free (an empty part has lse = -inf -> weight 0 and
        // o = 0) so that 8 independent 16-byte loads are in flight per thread
        int i = 0;
        for (; i + 8 <= p.n_parts; i += 8) {
            float l8[8];
            f32x4 o8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                l8[u] = ld1(parts_lse_ + (i + u) * part_lse_stride + lse_idx);
                o8[u] = ld4(parts_o_ + (i + u) * part_o_stride + o_idx);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float wgt = expf(l8[u] - mref);
                den += wgt;
                o += o8[u] * wgt;
            }
        }
        for (; i + 4 <= p.n_parts; i += 4) {
            float l4[4];
            f32x4 o4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                l4[u] = ld1(parts_lse_ + (i + u) * part_lse_stride + lse_idx);
                o4[u] = ld4(parts_o_ + (i + u) * part_o_stride + o_idx);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float wgt = expf(l4[u] - mref);
                den += wgt;
                o += o4[u] * wgt;
            }
        }
        for (; i < p.n_parts; ++i) {
            const float wgt = expf(ld1(parts_lse_ + i * part_lse_stride + lse_idx) - mref);
            den += wgt;
            o += ld4(parts_o_ + i * part_o_stride + o_idx) * wgt;
        }
    }
    if (joint && lnew != -INFINITY) {
        const float wgt = expf(lnew - mref);
        den += wgt;
        o += *reinterpret_cast<const f32x4*>(p.new_o + o_idx) * wgt;
    }
    float lse = -INFINITY;
    if (den > 0.f) {
        o = o / den;
        lse = mref + logf(den);
    }
    if (p.o32 && act) *reinterpret_cast<f32x4*>(p.o32 + o_idx) = o;
    // LS_NEW_DRAFT: the log-normaliser asked for is the tree kernel's L (triton_tree_attn.attention returns (o, L))
    if (p.lse_out && act && (tid & 31) == 0) p.lse_out[lse_idx] = (p.mode == LS_NEW_DRAFT) ? lnew : lse;
    if (XM == 1 && act) {
        // this rank's record, straight into every peer's mailbox (its own included): peer stores over xGMI
        const long slot = ((long)parity * p.x_world + p.x_rank) * p.x_cap;
        for (int dst = 0; dst < p.x_world; ++dst) {
            float* rec = reinterpret_cast<float*>(p.x_peers[dst] + XCHG_DATA_OFF) + slot;
            xchg_store16(rec + o_idx, o);
            if ((tid & 31) == 0) xchg_store4(rec + p.x_lse_off + lse_idx, lse);
        }
        xchg_stores_done();                        // acknowledged by every peer before this workgroup counts itself in
    }
    if constexpr (XM == 1) {
        __syncthreads();
        if (tid == 0) {
            unsigned int* ctr = &p.x_ctl->arrive_all;
            // RELAXED: the record travels as write-through stores that are acknowledged (vmcnt(0)) in front of the barrier above, so
            // the count has nothing to release -- an acq_rel count is `buffer_wbl2 sc1` + `buffer_inv sc1` in each of the 320
            // workgroups (round 4 A/B at 16k rows per rank: 26.3 -> 23.7 us per exchange, profiles/r4_xchg_arrive_relaxed.json)
            const unsigned prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x * gridDim.y * gridDim.z - 1) {       // the whole record has landed everywhere: raise the flags
                __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int dst = 0; dst < p.x_world; ++dst) xchg_raise_flag(p.x_peers[dst], parity, p.x_rank, epoch);
                __hip_atomic_store(&p.x_ctl->epoch, epoch + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!p.out || !act) return;

    float res[4];
    if (p.mode == LS_NEW_TARGET) {
        // prefix_o * weight + current_out * (1 - weight), every operand and result in the
        // activation dtype (llama.py:387; weight -> fp16 :420)
        const f32x4 cur = *reinterpret_cast<const f32x4*>(p.new_o + o_idx);
        const float wt = round_to<E>(sigmoidf_(lse - lnew));
        const float omw = round_to<E>(1.0f - wt);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = round_to<E>(round_to<E>(o[e]) * wt);
            const float bb = round_to<E>(cur[e] * omw);
            res[e] = a + bb;
        }
    } else if (p.mode == LS_NEW_DRAFT) {
        // prefix_o.float() * weight + current_out * (1 - weight) in fp32 (llama_glide.py:302,326)
        const f32x4 cur = *reinterpret_cast<const f32x4*>(p.new_o + o_idx);
        const float wt = sigmoidf_(lse - lnew);
#pragma unroll
        for (int e = 0; e < 4; ++e) res[e] = round_to<E>(o[e]) * wt + cur[e] * (1.0f - wt);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) res[e] = o[e];
    }
    typename E::V4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = E::from_f32(res[e]);
    typename E::T* op = reinterpret_cast<typename E::T*>(p.out) + (long)bi * p.out_sb + (long)r * p.out_ss +
                        (long)h * p.out_sh + d;
    *reinterpret_cast<typename E::V4*>(op) = ov;
}

__global__ void pack_mask_kernel(const int64_t* mask, int M, int N, uint32_t* bits, int words, int total_rows) {
    const int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= total_rows) return;
    const int64_t* mr = mask + (long)row * N;
    for (int wd = 0; wd < words; wd += 2) {
        const int j = wd * 32 + lane;
        const bool v = (j < N) && (mr[j] != 0);
        const unsigned long long bal = __ballot(v);
        if (lane == 0) {
            bits[(long)row * words + wd] = (uint32_t)bal;
            if (wd + 1 < words) bits[(long)row * words + wd + 1] = (uint32_t)(bal >> 32);
        }
    }
}

// ---- host side ---------------------------------------------------------------------------
struct Cfg {
    int qtA, qtB, rbA, RB, KS, tile, bpw, nstages, nd, pp, row_chunks, rows_per_chunk, threads, lds;
    int ws;        // 1: warp-specialised prefix path (attn_partial_ws_kernel), 2: ping-pong (attn_partial_pp_kernel)
    int pp_extra;
    int ws_qt;     // row tiles per S/O pair of the warp-specialised kernel
};

// Workgroup shape for M = g*sq rows sharing one K/V stream (see the header comment).
// The warp-specialised kernel takes the verification-sized row blocks (17..20 tiles) of calls whose prefix has
// no causal / window edge (every row sees keys [0, L)) and whose new block fits its smaller ring.
// Which streaming kernel serves verification-sized row blocks: LS_ATTN_KERNEL = ws (default) | general, read ONCE per
// process (A/B switch of the benchmarks).
int kernel_choice() {       // 0 general, 1 warp-specialised (round 2), 2 ping-pong (round 3)
    static const int choice = [] {
        const char* e = getenv("LS_ATTN_KERNEL");
        if (e && e[0] == 'g') return 0;
#ifdef LS_WITH_W4
        if (e && e[0] == 'f') return 3;          // four-wave kernel (round 3, measured negative)
#endif
#ifdef LS_WITH_PP
        if (e && e[0] == 'p') return 2;
#endif
        return 1;
    }();
    return choice;
}

bool ws_eligible(const ls_attn_desc* d) {
    if (kernel_choice() < 1) return false;
    // causal with ALL query rows appended (the chunks of a long prompt, ops._prefill_attention_batched): the diagonal lives
    // in the new-key block and every row sees the whole prefix -- hi(r) = min(L, r + sk - sq + 1) = L when sk = L + sq.
    // Only for long prompts: short ones keep the kernel (and the rounding) the goldens were generated against.
    const bool append_chunk = d->causal != 0 && d->new_mode == LS_NEW_FLASH && d->n_app == d->sq && d->kv_len_hint >= 4096;
    int cap = WS_NEW_CAP;
#ifdef LS_WITH_PP
    if (kernel_choice() == 2) cap = PP_NEW_CAP;
#endif
#ifdef LS_WITH_W4
    if (kernel_choice() == 3) cap = W4_NEW_CAP;
#endif
    return (d->causal == 0 || append_chunk) && d->window_left < 0 && (d->new_mode == LS_NEW_NONE || d->n_new <= cap / 64 * 64);
}

Cfg pick_cfg(int M, bool ws_ok, bool long_prefix) {
    Cfg c;
    c.ws = 0;
    c.pp_extra = 0;
    c.ws_qt = WS_QT;
    int tiles = (M + 15) / 16;
    c.row_chunks = 1;
    if (tiles > 24) {                   // g*sq > 384 rows: several row chunks re-read the K/V stream
        c.row_chunks = (M + 319) / 320;
        tiles = 20;
    }
    const bool ws2 = ws_ok && long_prefix && kernel_choice() == 1 && tiles > 20 && tiles <= 24;
    // 21..24 row tiles (GQA-5 x 74 verification rows = 370: QwQ) run as TWO row chunks of 12 tiles on the warp-specialised
    // kernel with 3 tiles per S/O pair: six tiles per pair do not fit the S wave's registers (96 for Q^T alone), and the general
    // kernel that served this shape until round 3 ran it at 0.23 of the HBM roofline.  The chunks re-read the K/V stream (from
    // L2: the two workgroups of a split run side by side); half as many splits, the same partial volume.  Prefixes of 4096 rows
    // and more only: short ones keep the kernel (and the roundings) the bf16 golden runs were generated against -- a fixed
    // soft-max reference and bf16 row sums moved one token of `qwen_bf16_g5` (tests/test_gpu_generate.py).
    if (tiles <= 1) { c.qtA = c.qtB = 1; c.RB = 1; c.KS = 2; }
    else if (tiles <= 8) { c.qtA = c.qtB = 2; c.RB = (tiles + 1) / 2; c.KS = 2; }
    else if (tiles <= 16) { c.qtA = c.qtB = 2; c.RB = (tiles + 1) / 2; c.KS = 1; }
    else if (tiles <= 20) { c.qtA = 3; c.qtB = 2; c.RB = 8; c.KS = 1; }      // 3,3,3,3,2,2,2,2
    else { c.qtA = c.qtB = 3; c.RB = 8; c.KS = 1; }
    if (ws2) { c.row_chunks = 2; c.qtA = c.qtB = 3; c.RB = 4; c.KS = 1; }      // new-key block: 4 workers x 3 tiles per chunk
    c.rbA = c.qtA == c.qtB ? c.RB : 4;
    c.bpw = c.KS == 1 ? 2 : 1;          // 32-key blocks per wave and tile
    c.tile = 64;
    c.nstages = 4;                      // 128 KB LDS ring; 2 tiles (64 KB) in flight beside the 2 being read
    const int nw = ws2 ? 8 : c.RB * c.KS;
    c.nd = nw >= 8 ? 8 : (nw >= 4 ? 4 : (nw >= 2 ? 2 : 1));
    c.pp = (c.tile / 2) / c.nd;         // pieces (1 KB) per DMA wave and tile
    c.rows_per_chunk = (c.rbA * c.qtA + (c.RB - c.rbA) * c.qtB) * 16;
    c.threads = nw * 64;
    c.lds = c.nstages * 2 * c.tile * ROWB + 16;
#ifdef LS_WITH_PP
    if (ws_ok && kernel_choice() == 2 && tiles > 16 && tiles <= 24 && c.row_chunks == 1) {
        c.ws = 2;                                   // ping-pong kernel: 8 waves, 3 or 2 row tiles each
        c.nstages = PP_NEW_CAP / c.tile;            // the new-block workgroup's capacity in this ring
        c.lds = PP_LDS;
        c.pp_extra = tiles - 16;
    } else
#endif
#ifdef LS_WITH_W4
    if (ws_ok && kernel_choice() == 3 && tiles > 16 && tiles <= 20 && c.row_chunks == 1) {
        c.ws = 3;                                   // four-wave kernel: 4 waves x 5 row tiles
        c.qtA = c.qtB = 5; c.RB = 4; c.rbA = 4; c.KS = 1;
        c.rows_per_chunk = 320;
        c.threads = 256;
        c.nd = 4;
        c.nstages = W4_NEW_CAP / c.tile;
        c.lds = W4_LDS;
    } else
#endif
    if (ws2) {
        c.ws = 1;
        c.ws_qt = 3;
        c.nstages = WS_NEW_CAP / c.tile;
        c.lds = WS_LDS;
    } else if (ws_ok && tiles > 16 && tiles <= 20) {       // row split of the new-key block stays 3,3,3,3,2,2,2,2
        c.ws = 1;
        c.nstages = WS_NEW_CAP / c.tile;            // the new-block workgroup's capacity in the smaller ring
        c.lds = WS_LDS;
    }
    return c;
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

int pick_splits(const ls_attn_desc* d, const Cfg& c) {
    if (d->n_splits > 0) return d->n_splits;
    const int tile = c.tile;
    int span = d->kv_len_hint;
    if (d->window_left >= 0) span = span < d->window_left + d->sq + 1 ? span : d->window_left + d->sq + 1;
    int tiles = (span + tile - 1) / tile + (d->window_left >= 0 ? 1 : 0);
    if (tiles < 1) tiles = 1;
    const int wg_per_split = d->Hkv * c.row_chunks * d->b;
    // one workgroup per CU in total: the new-block workgroups (one per kv head) count too, otherwise the
    // surplus prefix workgroups start only when a CU frees up and the launch takes two rounds
    int target = num_cus() / wg_per_split - (d->new_mode != LS_NEW_NONE ? 1 : 0);
    if (target < 1) target = 1;
    int s = target < tiles ? target : tiles;
    if (s > 512) s = 512;
    // Splits whose first rows lie a multiple of 512 cache rows (1 MB at 8 kv heads) apart walk the same HBM channels in step
    // all launch long: 32 splits of a 131072-row prefix (4096 rows each) take 116 us per call, 31 splits 106.5
    // (tools/sweep_cross_attn_128k.py).  Calls with a new-key block have an odd split count already (one CU per kv head
    // goes to that block); the others give up a split for an odd stride.
    // The stride is computed the way the dispatched kernel cuts its splits (the warp-specialised kernel: ceil(blocks / s)
    // 32-key blocks from the hinted length; the general one: ceil(tiles / s) tiles), only for calls WITHOUT a new-key block,
    // and never below half the target (tiles = 16, s = 2 used to collapse to one split).
    if (d->new_mode == LS_NEW_NONE) {
        const int unit = c.ws ? 32 : tile;
        const int units = (span + unit - 1) / unit;
        int t = s;
        while (t > 1 && units > t && (((units + t - 1) / t) * unit) % 512 == 0) --t;
        if (2 * t >= s) s = t;
    }
    return s;
}

struct WsLayout {
    size_t parts_o, parts_lse, new_o, new_lse, total;
    int n_parts;
};

WsLayout ws_layout(const ls_attn_desc* d, const Cfg& c, int n_splits) {
    WsLayout w;
    w.n_parts = n_splits * c.KS;
    const size_t rows = (size_t)d->b * d->sq * d->H;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    w.parts_o = take((size_t)w.n_parts * rows * D * 4);
    w.parts_lse = take((size_t)w.n_parts * rows * 4);
    w.new_o = take(rows * D * 4);
    w.new_lse = take(rows * 4);
    w.total = off;
    return w;
}

int validate(const ls_attn_desc* d) {
    if (!d) LS_FAIL(LS_ERR_INVALID_ARG, "null descriptor");
    if (d->b < 1 || d->sq < 1 || d->H < 1 || d->Hkv < 1 || d->H % d->Hkv != 0)
        LS_FAIL(LS_ERR_INVALID_ARG, "bad dims b=%d sq=%d H=%d Hkv=%d", d->b, d->sq, d->H, d->Hkv);
    if (d->dtype != LS_F16 && d->dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "dtype %d", d->dtype);
    if (d->new_mode < LS_NEW_NONE || d->new_mode > LS_NEW_DRAFT) LS_FAIL(LS_ERR_INVALID_ARG, "new_mode %d", d->new_mode);
    if (d->new_mode != LS_NEW_NONE) {
        if (d->n_new < 1 || !d->mask_bits || d->mask_words * 32 < d->n_new || d->n_new_cached < 0 ||
            d->n_new_cached > d->n_new)
            LS_FAIL(LS_ERR_INVALID_ARG, "new block: n_new=%d cached=%d mask_words=%d", d->n_new, d->n_new_cached, d->mask_words);
        if (d->n_new_cached < d->n_new && (!d->k_new || !d->v_new)) LS_FAIL(LS_ERR_INVALID_ARG, "k_new/v_new missing");
    }
    if (d->new_mode != LS_NEW_NONE && d->n_new > 256) LS_FAIL(LS_ERR_UNSUPPORTED, "new block of %d keys > 256", d->n_new);
    if (!d->q || !d->k_cache || !d->v_cache || !d->cache_seqlens) LS_FAIL(LS_ERR_INVALID_ARG, "null tensor");
    if (d->kv_len_hint < 0) LS_FAIL(LS_ERR_INVALID_ARG, "kv_len_hint < 0");
    if ((d->q_stride_s | d->q_stride_h | d->kc_stride_s | d->kc_stride_h) & 7)
        LS_FAIL(LS_ERR_INVALID_ARG, "row strides must be multiples of 8 elements (16-byte vector loads)");
    return LS_OK;
}

template <typename E, int QTA, int QTB>
int launch_partial(const Cfg& c, const AttnK& k, dim3 grid, hipStream_t s) {
    auto fn = attn_partial_kernel<E, QTA, QTB>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr;                     // function-local static: initialised once, thread-safe
    hipLaunchKernelGGL(fn, grid, dim3(c.threads), c.lds, s, k);
    LS_CHECK_LAUNCH("attn_partial_kernel");
    return LS_OK;
}

template <typename E, int QT>
int launch_partial_ws(const Cfg& c, const AttnK& k, dim3 grid, hipStream_t s) {
    auto fn = attn_partial_ws_kernel<E, QT>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr;
    hipLaunchKernelGGL(fn, grid, dim3(c.threads), c.lds, s, k);
    LS_CHECK_LAUNCH("attn_partial_ws_kernel");
    return LS_OK;
}

#ifdef LS_WITH_PP
template <typename E>
int launch_partial_pp(const Cfg& c, const AttnK& k, dim3 grid, hipStream_t s) {
    auto fn = attn_partial_pp_kernel<E>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr;
    hipLaunchKernelGGL(fn, grid, dim3(c.threads), c.lds, s, k);
    LS_CHECK_LAUNCH("attn_partial_pp_kernel");
    return LS_OK;
}
#endif

#ifdef LS_WITH_W4
template <typename E>
int launch_partial_w4(const Cfg& c, const AttnK& k, dim3 grid, hipStream_t s) {
    auto fn = attn_partial_w4_kernel<E>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr;
    hipLaunchKernelGGL(fn, grid, dim3(c.threads), c.lds, s, k);
    LS_CHECK_LAUNCH("attn_partial_w4_kernel");
    return LS_OK;
}
#endif

template <typename E>
int dispatch_partial(const Cfg& c, const AttnK& k, dim3 grid, hipStream_t s) {
#ifdef LS_WITH_W4
    if (c.ws == 3) return launch_partial_w4<E>(c, k, grid, s);
#endif
#ifdef LS_WITH_PP
    if (c.ws == 2) return launch_partial_pp<E>(c, k, grid, s);
#endif
    if (c.ws) return c.ws_qt == 3 ? launch_partial_ws<E, 3>(c, k, grid, s) : launch_partial_ws<E, WS_QT>(c, k, grid, s);
    if (c.qtA == 1) return launch_partial<E, 1, 1>(c, k, grid, s);
    if (c.qtA == 2) return launch_partial<E, 2, 2>(c, k, grid, s);
    if (c.qtB == 2) return launch_partial<E, 3, 2>(c, k, grid, s);
    return launch_partial<E, 3, 3>(c, k, grid, s);
}

int run_partial(const ls_attn_desc* d, void* ws, size_t ws_bytes, hipStream_t s, WsLayout* out_layout) {
    int rc = validate(d);
    if (rc) return rc;
    const int g = d->H / d->Hkv;
    const Cfg c = pick_cfg(g * d->sq, ws_eligible(d), d->kv_len_hint >= 4096);
    const int n_splits = pick_splits(d, c);
    const WsLayout w = ws_layout(d, c, n_splits);
    if (!ws || ws_bytes < w.total) LS_FAIL(LS_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, w.total);
    AttnK k;
    k.q = d->q;
    k.k_cache = d->k_cache;
    k.v_cache = d->v_cache;
    k.k_cache_w = d->k_cache;
    k.v_cache_w = d->v_cache;
    k.k_new = d->k_new;
    k.v_new = d->v_new;
    k.cache_seqlens = d->cache_seqlens;
    k.mask_bits = d->mask_bits;
    char* base = static_cast<char*>(ws);
    k.parts_o = reinterpret_cast<float*>(base + w.parts_o);
    k.parts_lse = reinterpret_cast<float*>(base + w.parts_lse);
    k.new_o = reinterpret_cast<float*>(base + w.new_o);
    k.new_lse = reinterpret_cast<float*>(base + w.new_lse);
    k.b = d->b; k.sq = d->sq; k.H = d->H; k.Hkv = d->Hkv; k.g = g; k.M = g * d->sq;
    k.has_new = d->new_mode != LS_NEW_NONE;
    k.new_mode = d->new_mode;
    k.n_new = d->n_new;
    k.n_new_cached = d->n_new_cached;
    k.mask_words = d->mask_words;
    k.scatter_new = d->scatter_new;
    k.prescale_q = d->prescale_q;
    k.causal = d->causal;
    k.window_left = d->window_left;
    k.n_app = d->n_app;
    k.n_splits = n_splits;
    k.row_chunks = c.row_chunks;
    k.rows_per_chunk = c.rows_per_chunk;
    k.RB = c.RB; k.KS = c.KS; k.tile = c.tile; k.bpw = c.bpw; k.nstages = c.nstages; k.nd = c.nd; k.pp = c.pp;
    k.rbA = c.rbA; k.qtA = c.qtA; k.qtB = c.qtB;
    k.pp_extra = c.pp_extra;
    k.scale = d->softmax_scale;
    k.q_sb = d->q_stride_b; k.q_ss = d->q_stride_s; k.q_sh = d->q_stride_h;
    k.kc_sb = d->kc_stride_b; k.kc_ss = d->kc_stride_s; k.kc_sh = d->kc_stride_h;
    k.kn_sb = d->kn_stride_b; k.kn_ss = d->kn_stride_s; k.kn_sh = d->kn_stride_h;
    dim3 grid(n_splits + k.has_new, d->Hkv * c.row_chunks, d->b);
    if (d->ev_start) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_start), s);
    rc = d->dtype == LS_F16 ? dispatch_partial<ElemF16>(c, k, grid, s) : dispatch_partial<ElemBF16>(c, k, grid, s);
    if (d->ev_stop) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_stop), s);
    if (out_layout) *out_layout = w;
    return rc;
}

int run_finish(const ls_attn_desc* d, const float* parts_o, const float* parts_lse, int n_parts, const float* new_o,
               const float* new_lse, int mode, void* out, float* o32, float* lse, hipStream_t s, long part_o_stride = 0,
               long part_lse_stride = 0, const ls_xchg* x = nullptr, int x_mode = 0) {
    FinK f;
    f.x_mode = x ? x_mode : 0;
    f.x_rank = x ? x->rank : 0;
    f.x_world = x ? x->world : 0;
    f.x_cap = x ? (long)x->cap_floats : 0;
    f.x_lse_off = (long)d->b * d->sq * d->H * D;
    f.x_peers = x ? x->peers_dev : nullptr;
    f.x_box = x ? x->box : nullptr;
    f.x_ctl = x ? x->ctl : nullptr;
    f.part_o_stride = part_o_stride ? part_o_stride : (long)d->b * d->sq * d->H * D;
    f.part_lse_stride = part_lse_stride ? part_lse_stride : (long)d->b * d->H * d->sq;
    f.parts_o = parts_o;
    f.parts_lse = parts_lse;
    f.new_o = new_o;
    f.new_lse = new_lse;
    f.out = out;
    f.o32 = o32;
    f.lse_out = lse;
    f.n_parts = n_parts;
    f.b = d->b; f.sq = d->sq; f.H = d->H;
    f.mode = mode;
    f.out_sb = d->out_stride_b; f.out_ss = d->out_stride_s; f.out_sh = d->out_stride_h;
    dim3 grid((d->sq + 7) / 8, d->H, d->b);
    auto launch = [&](auto xm) {
        constexpr int XM = decltype(xm)::value;
        if (d->dtype == LS_F16)
            hipLaunchKernelGGL((attn_finish_kernel<ElemF16, XM>), grid, dim3(256), 0, s, f);
        else
            hipLaunchKernelGGL((attn_finish_kernel<ElemBF16, XM>), grid, dim3(256), 0, s, f);
    };
    if (f.x_mode == 1) launch(std::integral_constant<int, 1>{});
    else if (f.x_mode == 2) launch(std::integral_constant<int, 2>{});
    else launch(std::integral_constant<int, 0>{});
    LS_CHECK_LAUNCH("attn_finish_kernel");
    return LS_OK;
}

}  // namespace

extern "C" {

size_t ls_attn_workspace_bytes(const ls_attn_desc* d) {
    if (validate(d)) return 0;
    const Cfg c = pick_cfg(d->H / d->Hkv * d->sq, ws_eligible(d), d->kv_len_hint >= 4096);
    return ws_layout(d, c, pick_splits(d, c)).total;
}

int ls_attn_num_parts(const ls_attn_desc* d) {
    if (validate(d)) return LS_ERR_INVALID_ARG;
    const Cfg c = pick_cfg(d->H / d->Hkv * d->sq, ws_eligible(d), d->kv_len_hint >= 4096);
    return pick_splits(d, c) * c.KS;
}

const char* ls_attn_kernel_name(const ls_attn_desc* d) {
    if (validate(d)) return "invalid";
    const Cfg c = pick_cfg(d->H / d->Hkv * d->sq, ws_eligible(d), d->kv_len_hint >= 4096);
    return c.ws == 3 ? "attn_partial_w4_kernel" : c.ws == 2 ? "attn_partial_pp_kernel" : c.ws ? "attn_partial_ws_kernel" : "attn_partial_kernel";
}

long ls_attn_redo_count(int reset) {
    // synchronises the device: a diagnostic, never called on the decode path
    unsigned int v = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_attn_redo_count), sizeof(v)) != hipSuccess) return -1;
    if (reset) {
        const unsigned int z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_redo_count), &z, sizeof(z)) != hipSuccess) return -1;
    }
    return (long)v;
}

int ls_attn_partial(const ls_attn_desc* d, void* ws, size_t ws_bytes, void* stream) {
    return run_partial(d, ws, ws_bytes, static_cast<hipStream_t>(stream), nullptr);
}

int ls_attn_fwd(const ls_attn_desc* d, void* ws, size_t ws_bytes, void* stream) {
    WsLayout w;
    int rc = run_partial(d, ws, ws_bytes, static_cast<hipStream_t>(stream), &w);
    if (rc) return rc;
    if (!d->out) LS_FAIL(LS_ERR_INVALID_ARG, "out is null");
    if (d->lse && d->new_mode == LS_NEW_TARGET) LS_FAIL(LS_ERR_INVALID_ARG, "no lse output in LS_NEW_TARGET mode");
    char* base = static_cast<char*>(ws);
    const bool has_new = d->new_mode != LS_NEW_NONE;
    return run_finish(d, reinterpret_cast<float*>(base + w.parts_o), reinterpret_cast<float*>(base + w.parts_lse),
                      w.n_parts, has_new ? reinterpret_cast<float*>(base + w.new_o) : nullptr,
                      has_new ? reinterpret_cast<float*>(base + w.new_lse) : nullptr, d->new_mode, d->out, nullptr,
                      d->lse, static_cast<hipStream_t>(stream));
}

int ls_attn_reduce_local(const ls_attn_desc* d, void* ws, size_t ws_bytes, float* o32, float* lse, void* stream) {
    if (validate(d)) return LS_ERR_INVALID_ARG;
    if (!o32 || !lse) LS_FAIL(LS_ERR_INVALID_ARG, "o32/lse null");
    const Cfg c = pick_cfg(d->H / d->Hkv * d->sq, ws_eligible(d), d->kv_len_hint >= 4096);
    const WsLayout w = ws_layout(d, c, pick_splits(d, c));
    if (!ws || ws_bytes < w.total) LS_FAIL(LS_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, w.total);
    char* base = static_cast<char*>(ws);
    return run_finish(d, reinterpret_cast<float*>(base + w.parts_o), reinterpret_cast<float*>(base + w.parts_lse),
                      w.n_parts, nullptr, nullptr, LS_NEW_NONE, nullptr, o32, lse, static_cast<hipStream_t>(stream));
}

int ls_attn_finish(const ls_attn_desc* d, const float* parts_o, const float* parts_lse, int n_parts, int64_t part_o_stride,
                   int64_t part_lse_stride, void* ws, size_t ws_bytes, void* stream) {
    if (validate(d)) return LS_ERR_INVALID_ARG;
    if (!parts_o || !parts_lse || n_parts < 1 || !d->out) LS_FAIL(LS_ERR_INVALID_ARG, "parts/out null");
    const Cfg c = pick_cfg(d->H / d->Hkv * d->sq, ws_eligible(d), d->kv_len_hint >= 4096);
    const WsLayout w = ws_layout(d, c, pick_splits(d, c));
    const bool has_new = d->new_mode != LS_NEW_NONE;
    if (has_new && (!ws || ws_bytes < w.total)) LS_FAIL(LS_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, w.total);
    char* base = static_cast<char*>(ws);
    return run_finish(d, parts_o, parts_lse, n_parts, has_new ? reinterpret_cast<float*>(base + w.new_o) : nullptr,
                      has_new ? reinterpret_cast<float*>(base + w.new_lse) : nullptr, d->new_mode, d->out, nullptr,
                      d->lse, static_cast<hipStream_t>(stream), (long)part_o_stride, (long)part_lse_stride);
}

static int xchg_fits(const ls_attn_desc* d, const ls_xchg* x) {
    if (!x || !x->connected) LS_FAIL(LS_ERR_INVALID_ARG, "exchange object null or not connected");
    const size_t rec = (size_t)d->b * d->sq * d->H * (D + 1);
    if (rec > x->cap_floats) LS_FAIL(LS_ERR_INVALID_ARG, "record of %zu floats exceeds the mailbox slot (%zu)", rec, x->cap_floats);
    return LS_OK;
}

int ls_attn_reduce_push(const ls_attn_desc* d, void* ws, size_t ws_bytes, ls_xchg* x, void* stream) {
    if (validate(d)) return LS_ERR_INVALID_ARG;
    if (xchg_fits(d, x)) return LS_ERR_INVALID_ARG;
    const Cfg c = pick_cfg(d->H / d->Hkv * d->sq, ws_eligible(d), d->kv_len_hint >= 4096);
    const WsLayout w = ws_layout(d, c, pick_splits(d, c));
    if (!ws || ws_bytes < w.total) LS_FAIL(LS_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, w.total);
    char* base = static_cast<char*>(ws);
    return run_finish(d, reinterpret_cast<float*>(base + w.parts_o), reinterpret_cast<float*>(base + w.parts_lse), w.n_parts, nullptr,
                      nullptr, LS_NEW_NONE, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream), 0, 0, x, 1);
}

int ls_attn_finish_xchg(const ls_attn_desc* d, ls_xchg* x, void* ws, size_t ws_bytes, void* stream) {
    if (validate(d)) return LS_ERR_INVALID_ARG;
    if (!d->out) LS_FAIL(LS_ERR_INVALID_ARG, "out null");
    if (xchg_fits(d, x)) return LS_ERR_INVALID_ARG;
    const Cfg c = pick_cfg(d->H / d->Hkv * d->sq, ws_eligible(d), d->kv_len_hint >= 4096);
    const WsLayout w = ws_layout(d, c, pick_splits(d, c));
    const bool has_new = d->new_mode != LS_NEW_NONE;
    if (has_new && (!ws || ws_bytes < w.total)) LS_FAIL(LS_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, w.total);
    char* base = static_cast<char*>(ws);
    // parts = the mailbox slots of this epoch (resolved on the device); stride = one slot
    return run_finish(d, nullptr, nullptr, x->world, has_new ? reinterpret_cast<float*>(base + w.new_o) : nullptr,
                      has_new ? reinterpret_cast<float*>(base + w.new_lse) : nullptr, d->new_mode, d->out, nullptr, d->lse,
                      static_cast<hipStream_t>(stream), (long)x->cap_floats, (long)x->cap_floats, x, 2);
}

int ls_lse_merge(const float* parts_o, const float* parts_lse, int n_parts, int b, int sq, int H, int dtype, void* out,
                 float* o32, float* lse, void* stream) {
    if (!parts_o || !parts_lse || n_parts < 1 || b < 1 || sq < 1 || H < 1) LS_FAIL(LS_ERR_INVALID_ARG, "bad merge args");
    ls_attn_desc d = {};
    d.b = b; d.sq = sq; d.H = H; d.dtype = dtype;
    d.out_stride_h = D; d.out_stride_s = (int64_t)H * D; d.out_stride_b = (int64_t)sq * H * D;
    return run_finish(&d, parts_o, parts_lse, n_parts, nullptr, nullptr, LS_NEW_NONE, out, o32, lse,
                      static_cast<hipStream_t>(stream));
}

int ls_pack_tree_mask(const int64_t* tree_mask, int b, int M, int N, uint32_t* bits, int words, void* stream) {
    if (!tree_mask || !bits || b < 1 || M < 1 || N < 1 || words * 32 < N) LS_FAIL(LS_ERR_INVALID_ARG, "bad mask args");
    const int rows = b * M;
    hipLaunchKernelGGL(pack_mask_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       tree_mask, M, N, bits, words, rows);
    LS_CHECK_LAUNCH("pack_mask_kernel");
    return LS_OK;
}

}  // extern "C"
