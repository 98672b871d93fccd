// gemm.hip -- weight-streaming "skinny" linear layers for the decode round (gfx950 / CDNA4).
//
// Every projection of a draft or verify pass multiplies a handful of token rows (M = 1 ... 80:
// 74 verification rows, 4/16 tree-level rows, 1 vanilla row) by a weight matrix that has to be read
// from HBM in full: y[M,N] = x[M,K] . W[N,K]^T.  The flops are negligible next to the bytes of W, so
// the kernel is organised around the weight stream, not around an output tile:
//
//   * Weights are PRE-PACKED once (ls_linear_pack_weight) into the A-operand layout of
//     v_mfma_f32_16x16x32: the 1 KB block of (16-row tile, k-step s) holds lane l's 16 bytes
//     W[row0 + (l&15)][32s + 8(l>>4) ...] at byte 16*l, and the 4 tiles of a 64-row slab are adjacent.
//     A wave-wide global_load_dwordx4 of a block is one fully coalesced 1 KB read (8 full 128 B lines)
//     that lands directly in MFMA operand registers: W never touches LDS, and a wave walking the
//     k-steps of its slab streams contiguous memory, 4 KB per k-step.  (Loading
//     the row-major nn.Linear layout in MFMA lane order instead costs 64 separate 16 B L1 accesses per
//     instruction -- measured: 4x the TCP accesses and 2x the L2 requests of hipBLASLt, ~2.5 TB/s.)
//   * x (the few token rows; L2 resident) is the B operand.  Each wave stages the 64-k chunk it is
//     about to multiply through a wave-private LDS slab: coalesced loads (8 lanes = one 128 B line of a
//     row), XOR-swizzled ds_write_b128, conflict-free ds_read_b128 in MFMA lane order.  No workgroup
//     barrier is involved: LDS operations of one wave execute in order.
//   * A wave owns 64 output columns (4 MFMA tiles) x one quarter of the workgroup's k-range, so every x
//     fragment feeds 4 MFMAs; the next chunk of W (and of x) is in flight while the current one is
//     multiplied.  The 4 waves are reduced through LDS in a fixed order.
//   * Split-K across workgroups (needed to fill 256 CUs when N/64 < 256) is reduced in the SAME launch,
//     deterministically: partials go to a workspace with agent-coherent (write-through) stores, the
//     last workgroup to arrive at the slab's counter sums all S partials in split order -- the result
//     does not depend on arrival order -- and runs the epilogue.  No L2 write-back / invalidate is needed.
//     The k order of a row's dot product depends on (N, K) only, never on M: a token gets bit-identical
//     logits whether it is verified in a 74-row pass or decoded alone (vanilla).
//
// Epilogues: bias (q/k/v of the draft layer and of Qwen2), up to three weight segments sharing one x
// (q|k|v in one launch), and silu(gate) * up for the MLP -- with the reference's rounding points kept:
// each linear's output is rounded to the storage dtype before the next op uses it.
//
// Replaces, on the decode path (reference = longspec/test): llama.py:361-363,390 (q/k/v/o_proj),
// LlamaMLP.forward (transformers; vendored qwen2.py:218-230), llama_glide.py:248-250,268,285-287,305
// (draft projections), lm_head at llama_glide.py:960,1019,1046,1091.
#include <math.h>
#include <stdlib.h>

#include "ls_common.h"
#ifdef LS_WITH_TAIL
#define LS_TAIL_PART 0
#include "../../tools/mb/layer_tail_kernel.inc"
#undef LS_TAIL_PART
#endif

namespace {

constexpr int GEMM_THREADS = 256;
// Bytes of padding between the 64-row groups of a packed weight (experiment hook, 0 in the product).  The groups lie
// nks * 4 KB apart -- 512 KB for K = 4096 -- and the workgroups of a launch walk them at the same pace; an 8 MB stride
// between the splits costs the attention 9 % (tools/sweep_cross_attn_128k.py).  Here it does not: 256 B, 4 KB and 36 KB of
// padding all measured within +-2 us of the unpadded layout on every projection (profiles/r3_gemm_group_pad.txt).
#ifndef LS_GEMM_GROUP_PAD
#define LS_GEMM_GROUP_PAD 0
#endif
constexpr long GROUP_PAD = LS_GEMM_GROUP_PAD;
constexpr int COUNTER_BYTES = 64 * 1024;   // fixed counter region at the head of the workspace (16384 slabs)

struct GemmK {
    const char* x;
    long ldx;                    // elements
    const char* w[3];            // segment weights [n_i, K]  (SILU: w[0] = gate, w[1] = up)
    const char* bias[3];         // or null
    int n[3];                    // segment rows
    char* y;
    long ldy;                    // elements
    float* part;                 // split-K partials
    unsigned* counters;          // one per slab, zero between launches
    int M, K, N;                 // N = total output columns
    int nks, S, nslabs;          // nks = K / 32 k-steps
    int flag_off;                // byte offset of the last-arriver flag in dynamic LDS
    const char* rope_cos;        // EPI_QKV_ROPE: [M, 128] dtype tables of the rows' positions
    const char* rope_sin;
    int rope_segs;               // leading segments (q, k) that are rotated; the rest (v) are plain
    const char* residual;        // EPI_NONE: [M, N] dtype added to the rounded output (`residual + mlp(x)`), or null
    long ldr;
    // RMSNorm folded into the launch (NORM): x is the UN-normalised residual stream; its rows' sums of squares arrive
    // as `ssq_parts` partials per row (one per 64 columns, written by the launch that produced x), the norm weight is
    // applied while x is staged.  `ssq_out` (EPI_NONE): this launch is such a producer.
    const float* ssq_in;         // [M, ssq_parts]
    int ssq_parts;
    const char* norm_w;          // [K] dtype
    float norm_eps;
    float* ssq_out;              // [M, N / 64] or null
    int prefetch_units;          // > 0: ls_linear_prefetch -- every wave only REQUESTS its first units (default cache policy) and exits
};

enum { EPI_NONE = 0, EPI_SILU_MUL = 1, EPI_QKV_ROPE = 2 };

// agent-coherent accesses (sc1: write-through / cache-bypassing), so that partials written by a workgroup
// on one XCD are read correctly by the reducing workgroup on another without an L2 write-back + invalidate
// 16 bytes per lane (one accumulator quad): scalar sc1 stores are one fabric write EACH -- a dword costs ~6x the time per
// byte of a dwordx4 (MI355X_MICROARCH, visibility table) -- so a partial tile travels as `buffer_store_dwordx4 ... sc1` of
// the lane's f32x4, 1 KB contiguous per wave-instruction.  aux 16 = sc1 on gfx950.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t part_rsrc(const float* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void st_coherent4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ f32x4 ld_coherent4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}

// The weight stream: every byte is read ONCE per launch by ONE workgroup, so it is requested non-temporally
// (`global_load_dwordx4 ... nt`: no allocation priority in L2 / MALL -- the x rows, the partials and the next kernel's
// operands keep the cache).  MI355X_MICROARCH "nt-weights": issue -> landed -18 %, 5-10 % per decode layer.
// -DLS_GEMM_NT=0 builds the default-policy variant for A/B runs (tools/build_variant.py).
#ifndef LS_GEMM_NT
#define LS_GEMM_NT 1
#endif
// -DLS_GEMM_KSTEP8=1: k-step-granular sets for the 128-row variant (24 KB per wave in flight instead of one 16 KB chunk of
// look-ahead).  Round-4 A/B on one box: lm_head 202 vs 207 us, gate|up+SiLU 47.0 vs 45.4 us, GEMM per round 5.57-5.72 vs
// 5.60-5.65 ms -- no gain: the 128-row launches are not limited by their look-ahead.  Default: the round-3 pipeline.
#ifndef LS_GEMM_KSTEP8
#define LS_GEMM_KSTEP8 0
#endif
template <typename V>
__device__ __forceinline__ V load_w(const char* p) {
#if LS_GEMM_NT
    return __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#else
    return *reinterpret_cast<const V*>(p);
#endif
}

// MT = 16-row tiles of x (M <= 16*MT); NT = 16-row weight tiles per workgroup (4: one packed slab, 8: two)
// NORM: `x` is normalised on the way into LDS -- LlamaRMSNorm's arithmetic (misc.hip::rmsnorm_rows_kernel), with the row
// sums of squares taken from the producer's 64-column partials in the canonical order (ls_common.h::ssq_*).
template <typename E, int MT, int NT, int EPI, bool NORM>
__global__ __launch_bounds__(GEMM_THREADS, MT >= 5 ? 1 : 2) void skinny_gemm_kernel(const GemmK p) {
    using V8 = typename E::V8;
    using V4 = typename E::V4;
    // chunks (2 k-steps = 64 k; NT x 2 KB of W) of look-ahead per wave.  M <= 32: 2 waves per SIMD (256
    // registers each); above: 1 wave per SIMD with the full 512.
    // A W register set ("unit") holds KPS k-steps of the workgroup's NT tiles: a whole 64-k chunk (2 k-steps x 4 tiles) for
    // the 64-row variants, ONE k-step x 8 tiles for the 128-row variant -- 32 registers either way, so that the 128-row
    // launches (gate|up, lm_head) keep three sets = 24 KB per wave in flight like the others instead of one 16 KB chunk
    // (round 4: they streamed at 5.2 TB/s against 5.9 for the 64-row launches of the same pass).
    constexpr int KPS = (NT == 8 && LS_GEMM_KSTEP8) ? 1 : 2;
    constexpr int UPC = 2 / KPS;             // units per 64-k chunk of x
    constexpr int LAC = NT == 8 ? (KPS == 1 ? 3 : 1) : MT == 1 ? 4 : 3;      // units of look-ahead
    constexpr int NCS = LAC + 1;             // W register sets, one per unit in flight
    static_assert(NCS % UPC == 0, "the x-chunk phase of a unit must be static inside the unrolled loop");
    constexpr int XL = 2 * MT;               // 1 KB pieces (8 rows x 128 B) of one x chunk
    constexpr int XSLAB = MT * 16 * 128;     // bytes of a wave's x slab
    constexpr int NPASS = NT / 4;            // the 4-wave reduction handles 4 tiles per pass (LDS budget)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int slab = blockIdx.x, split = blockIdx.y;
    const int nch_all = p.nks >> 1;          // 64-k chunks of K
    const int ch_begin = (int)(((long)nch_all * split) / p.S);
    const int ch_end = (int)(((long)nch_all * (split + 1)) / p.S);

    // ---- the NT weight tiles of this workgroup.  Packed block (64-row group g, k-step s, tile t) is the 1 KB
    // at ((g * nks + s) * 4 + t) * 1024.  EPI_SILU_MUL: the packed matrix alternates gate and up tiles (tile 2j =
    // gate rows 16j.., tile 2j+1 = up rows 16j..), so tiles (2j, 2j+1) make output columns 16j..16j+15.
    // EPI_QKV_ROPE: like EPI_NONE, but the q and k segments are packed by ls_linear_pack_rope (within each 128-row
    // head, tile 2j = rows 16j.., tile 2j+1 = rows 64+16j..: a rotary pair sits in neighbouring tiles).
    const char* wtile[NT];
    const char* bias_p = nullptr;
    int n_lim;                   // end of the valid output columns of this slab's segment (global column)
    int n_tile0;                 // global output column of tile 0
    int seg_base = 0, seg = 0;
    const long group_b = (long)p.nks * 4096 + GROUP_PAD;
    {
        const int row0 = slab * NT * 16;                 // first packed row of the workgroup (global over segments)
        if (EPI != EPI_SILU_MUL) {
            if (row0 >= p.n[0]) { seg_base = p.n[0]; seg = 1; }
            if (seg == 1 && row0 >= p.n[0] + p.n[1]) { seg_base = p.n[0] + p.n[1]; seg = 2; }
            bias_p = p.bias[seg] ? p.bias[seg] - (long)seg_base * 2 : nullptr;   // indexable by global column
            n_tile0 = row0;
            n_lim = seg_base + p.n[seg];
        } else {
            n_tile0 = row0 >> 1;
            n_lim = p.n[0];
        }
        const int ngroups = EPI != EPI_SILU_MUL ? (p.n[seg] + 63) >> 6 : (2 * p.n[0] + 63) >> 6;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int g = min(((row0 - seg_base) >> 6) + (t >> 2), ngroups - 1);    // clamp: tiles past the end are never stored
            wtile[t] = p.w[seg] + (long)g * group_b + (t & 3) * 1024 + lane * 16;
        }
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // wave w takes the w-th contiguous quarter of the workgroup's chunk range
    const int quarter = (ch_end - ch_begin + 3) >> 2;
    const int ch0 = ch_begin + wave * quarter;
    const int nch = max(0, min(ch_end - ch0, quarter));

    V8 wa[NCS][KPS][NT];         // [set][k-step in unit][tile]
    V8 xs[XL];                   // x staging: piece i = rows 8i .. 8i+7, lane -> (row 8i + lane/8, 16 B slot lane%8)
    char* xlds = smem + wave * XSLAB;
    const int xr_in = lane >> 3, xslot = lane & 7;

    unsigned xoff[XL];           // 32-bit byte offsets of this lane's 16 B in each piece (x is far below 4 GB)
#pragma unroll
    for (int i = 0; i < XL; ++i) xoff[i] = (unsigned)(((long)min(i * 8 + xr_in, p.M - 1) * p.ldx + xslot * 8) * 2);
    // NORM: behind the last-arriver flag the launch keeps 1 / rms of every row (fp32 [MT * 16]) and the norm weight
    // (dtype [K]) in LDS; both are read back per piece / per chunk rather than held in registers (the 128-row variant
    // has none to spare)
    const float* rs = reinterpret_cast<const float*>(smem + p.flag_off + 16);
    const char* nwl = smem + p.flag_off + 16 + MT * 64;
    int x_ch = 0;                // the chunk in `xs`
    auto load_x = [&](int ch) {
        const char* xc = p.x + (long)ch * 128;            // wave-uniform
#pragma unroll
        for (int i = 0; i < XL; ++i) xs[i] = *reinterpret_cast<const V8*>(xc + xoff[i]);
        if (NORM) x_ch = ch;
    };
    auto store_x = [&]() {       // slot ^ ((row >> 1) & 7): 16 rows x one slot hit 16 distinct 16 B bank groups
        V8 nw;
        if (NORM) nw = *reinterpret_cast<const V8*>(nwl + x_ch * 128 + xslot * 16);
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int row = i * 8 + xr_in;
            V8 v = xs[i];
            if (NORM) {
                const float rstd = rs[row];
#pragma unroll
                for (int e = 0; e < 8; ++e) {             // weight * dtype(x32 * rsqrt(mean(x32^2) + eps))
                    const float n = round_to<E>(E::to_f32(v[e]) * rstd);
                    v[e] = E::from_f32(E::to_f32(nw[e]) * n);
                }
            }
            *reinterpret_cast<V8*>(xlds + row * 128 + ((xslot ^ ((row >> 1) & 7)) << 4)) = v;
        }
    };
    // unit `un` of this wave = k-steps ch0 * 2 + un * KPS ... of the slab
    auto issue_w = [&](int un, int set) {
#pragma unroll
        for (int kk = 0; kk < KPS; ++kk)
#pragma unroll
            for (int t = 0; t < NT; ++t) wa[set][kk][t] = load_w<V8>(wtile[t] + (long)(ch0 * 2 + un * KPS + kk) * 4096);
    };
    // `ph` = the unit's position inside its x chunk (0 .. UPC - 1; static)
    auto mma_unit = [&](int ph, int set) {
#pragma unroll
        for (int kk = 0; kk < KPS; ++kk) {
            const int ks = ph * KPS + kk;                    // k-step inside the 64-k chunk staged in LDS
            V8 bx[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = mt * 16 + l15;
                bx[mt] = *reinterpret_cast<const V8*>(xlds + row * 128 + (((ks * 4 + g4) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[t][mt] = E::mfma(wa[set][kk][t], bx[mt], acc[t][mt]);
        }
    };

    const int nun = nch * UPC;                               // units of this wave
    if (p.prefetch_units > 0) {
        // ls_linear_prefetch: pull the first units of every wave's weight stream into the L2 of the XCD that the matching
        // workgroup of the real launch will run on (same grid, block b -> XCD b % 8 as observed; speed only), with the
        // DEFAULT cache policy -- the later nt loads hit those lines
        const int n = min(nun, p.prefetch_units);
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int kk = 0; kk < KPS; ++kk)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const V8 v = *reinterpret_cast<const V8*>(wtile[t] + (long)(ch0 * 2 + i * KPS + kk) * 4096);
                    asm volatile("" : : "v"(v));
                }
        return;
    }
    if (nch > 0) load_x(ch0);
#pragma unroll
    for (int i = 0; i < LAC; ++i)
        if (i < nun) issue_w(i, i);
    if (NORM) {
        // behind the first weight requests: thread r sums row r's partials in slab order (the canonical order)
        float* rsw = reinterpret_cast<float*>(smem + p.flag_off + 16);
        for (int i = tid * 16; i < p.K * 2; i += GEMM_THREADS * 16)
            *reinterpret_cast<uint4*>(smem + p.flag_off + 16 + MT * 64 + i) = *reinterpret_cast<const uint4*>(p.norm_w + i);
        if (tid < MT * 16) {
            const float tot = ssq_row(p.ssq_in + (long)min(tid, p.M - 1) * p.ssq_parts, p.ssq_parts);
            rsw[tid] = rsqrtf(tot / (float)p.K + p.norm_eps);
        }
        __syncthreads();
    }
    int c = 0;
    // steady state: no control flow inside, so the compiler's in-order vmcnt counts stay exact;
    // sched_barrier(0) keeps the loads of the chunks ahead in front of the MFMAs of the current one
    for (; c + NCS - 1 + LAC < nun; c += NCS) {            // `c` (a multiple of NCS, hence of UPC) counts units
#pragma unroll
        for (int u = 0; u < NCS; ++u) {
            if (u % UPC == 0) {
                store_x();                               // the chunk of unit c+u: staging registers -> the wave's LDS slab
                load_x(ch0 + (c + u) / UPC + 1);         // (the last chunk of the wave is never in this loop: LAC >= UPC)
            }
            issue_w(c + u + LAC, (u + LAC) % NCS);
            __builtin_amdgcn_sched_barrier(0);
            mma_unit(u % UPC, u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < NCS + LAC - 1; ++u) {            // drain
        if (c + u < nun) {
            if (u % UPC == 0) {
                store_x();
                if ((c + u) / UPC + 1 < nch) load_x(ch0 + (c + u) / UPC + 1);
            }
            if (c + u + LAC < nun) issue_w(c + u + LAC, (u + LAC) % NCS);
            __builtin_amdgcn_sched_barrier(0);
            mma_unit(u % UPC, u % NCS);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- reduce the 4 waves (fixed order) through LDS, 4 tiles per pass.  EPI_NONE: wave w finishes tile
    // 4h + w of pass h.  EPI_SILU_MUL / EPI_QKV_ROPE: waves 0,1 finish the tile pair 4h + 2w, 4h + 2w + 1
    // (gate, up) / (rotary low half, high half).
    constexpr bool PAIRED = EPI != EPI_NONE;
    constexpr int NT_OUT = PAIRED ? 2 : 1;                   // tiles per finishing wave per pass
    const bool finisher = PAIRED ? wave < 2 : true;
    float* red = reinterpret_cast<float*>(smem);
    f32x4 r[NPASS][NT_OUT][MT];
#pragma unroll
    for (int h = 0; h < NPASS; ++h) {
        __syncthreads();         // x slabs / previous pass no longer read
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                *reinterpret_cast<f32x4*>(red + (((wave * 4 + t) * MT + mt) * 64 + lane) * 4) = acc[h * 4 + t][mt];
        __syncthreads();
        if (finisher) {
#pragma unroll
            for (int q = 0; q < NT_OUT; ++q) {
                const int t = PAIRED ? 2 * wave + q : wave;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(red + (((0 * 4 + t) * MT + mt) * 64 + lane) * 4);
#pragma unroll
                    for (int w2 = 1; w2 < 4; ++w2)
                        v += *reinterpret_cast<const f32x4*>(red + (((w2 * 4 + t) * MT + mt) * 64 + lane) * 4);
                    r[h][q][mt] = v;
                }
            }
        }
    }

    // ---- split-K: deterministic last-arriver reduction
    if (p.S > 1) {
        constexpr int TILE_F = MT * 4 * 64;                  // floats of one tile's accumulators
        auto tile_of = [&](int h, int q) { return h * 4 + (PAIRED ? 2 * wave + q : wave); };
        if (finisher) {
#pragma unroll
            for (int h = 0; h < NPASS; ++h)
#pragma unroll
                for (int q = 0; q < NT_OUT; ++q) {
                    const __amdgpu_buffer_rsrc_t mine =
                        part_rsrc(p.part + (((long)split * p.nslabs + slab) * NT + tile_of(h, q)) * TILE_F, TILE_F * 4);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) st_coherent4(mine, (mt * 64 + lane) * 16, r[h][q][mt]);
                }
        }
        // Every thread drains its OWN write-through (sc1) partial stores before the barrier: only then may thread 0
        // bump the slab counter.  (A workgroup-scope fence does not emit the wait -- the compiler left vmcnt(63)
        // in front of the barrier -- so the last arriver, possibly on another XCD, could sum partials still in
        // flight.  The asm wait is invisible to the waitcnt-elision pass: MI355X_MICROARCH, "Compiler hazard".)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        volatile unsigned& s_last = *reinterpret_cast<volatile unsigned*>(smem + p.flag_off);
        if (tid == 0) {
            const unsigned prev = __hip_atomic_fetch_add(p.counters + slab, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (prev == (unsigned)p.S - 1u);
            if (prev == (unsigned)p.S - 1u) __hip_atomic_store(p.counters + slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;
        if (finisher) {
#pragma unroll
            for (int h = 0; h < NPASS; ++h)
#pragma unroll
                for (int q = 0; q < NT_OUT; ++q)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) r[h][q][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < p.S; ++s) {
#pragma unroll
                for (int h = 0; h < NPASS; ++h)
#pragma unroll
                    for (int q = 0; q < NT_OUT; ++q) {
                        const __amdgpu_buffer_rsrc_t src =
                            part_rsrc(p.part + (((long)s * p.nslabs + slab) * NT + tile_of(h, q)) * TILE_F, TILE_F * 4);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) r[h][q][mt] += ld_coherent4(src, (mt * 64 + lane) * 16);
                    }
            }
        }
    }
    if (!finisher) return;

    // ---- epilogue: lane holds y[m = mt*16 + l15][n .. n+3]
#pragma unroll
    for (int h = 0; h < NPASS; ++h) {
        if (EPI == EPI_SILU_MUL) {
            const int nn = n_tile0 + (h * 2 + wave) * 16 + g4 * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + l15;
                V4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = round_to<E>(r[h][0][mt][e]);              // gate_proj output in the storage dtype
                    const float u = round_to<E>(r[h][NT_OUT - 1][mt][e]);     // up_proj output
                    const float sg = round_to<E>(g / (1.0f + expf(-g)));      // act_fn (SiLU), fp32 math, rounded
                    o[e] = E::from_f32(sg * u);
                }
                if (m < p.M && nn < n_lim) *reinterpret_cast<V4*>(p.y + ((long)m * p.ldy + nn) * 2) = o;
            }
        } else if (EPI == EPI_QKV_ROPE) {
            // tiles (2u, 2u+1) of the slab, u = 2h + wave.  Rotated segment: they are rows d.. and 64+d.. of one head;
            // apply_rotary_pos_emb on the rounded projections exactly as rope_apply_kernel does (misc.hip).
            const int tl = ((n_tile0 - seg_base) >> 4) + (h * 2 + wave) * 2;          // packed tile index in the segment
            const bool rot = seg < p.rope_segs;
            const int d = ((tl & 7) >> 1) * 16 + g4 * 4;                              // dimension of the low half
            const int n_lo = rot ? seg_base + (tl >> 3) * 128 + d : seg_base + tl * 16 + g4 * 4;
            const int n_hi = rot ? n_lo + 64 : n_lo + 16;
            float bl[4] = {0.f, 0.f, 0.f, 0.f}, bh[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias_p != nullptr) {
                if (n_lo < n_lim) {
                    const V4 b4 = *reinterpret_cast<const V4*>(bias_p + (long)n_lo * 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bl[e] = E::to_f32(b4[e]);
                }
                if (n_hi < n_lim) {
                    const V4 b4 = *reinterpret_cast<const V4*>(bias_p + (long)n_hi * 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bh[e] = E::to_f32(b4[e]);
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + l15;
                if (m >= p.M) continue;
                V4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = E::from_f32(r[h][0][mt][e] + bl[e]);
                    hi[e] = E::from_f32(r[h][1][mt][e] + bh[e]);
                }
                if (rot) {
                    const V4 cl = *reinterpret_cast<const V4*>(p.rope_cos + ((long)m * 128 + d) * 2);
                    const V4 ch = *reinterpret_cast<const V4*>(p.rope_cos + ((long)m * 128 + 64 + d) * 2);
                    const V4 sl = *reinterpret_cast<const V4*>(p.rope_sin + ((long)m * 128 + d) * 2);
                    const V4 sh = *reinterpret_cast<const V4*>(p.rope_sin + ((long)m * 128 + 64 + d) * 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xl = E::to_f32(lo[e]), xh = E::to_f32(hi[e]);
                        lo[e] = E::from_f32(round_to<E>(xl * E::to_f32(cl[e])) + round_to<E>(-xh * E::to_f32(sl[e])));
                        hi[e] = E::from_f32(round_to<E>(xh * E::to_f32(ch[e])) + round_to<E>(xl * E::to_f32(sh[e])));
                    }
                }
                if (n_lo < n_lim) *reinterpret_cast<V4*>(p.y + ((long)m * p.ldy + n_lo) * 2) = lo;
                if (n_hi < n_lim) *reinterpret_cast<V4*>(p.y + ((long)m * p.ldy + n_hi) * 2) = hi;
            }
        } else {
            const int nn = n_tile0 + (h * 4 + wave) * 16 + g4 * 4;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            float tsq[MT];
            if (bias_p != nullptr && nn < n_lim) {
                const V4 b4 = *reinterpret_cast<const V4*>(bias_p + (long)nn * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = E::to_f32(b4[e]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + l15;
                V4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = E::from_f32(r[h][0][mt][e] + bv[e]);
                if (m < p.M && nn < n_lim) {
                    if (p.residual != nullptr) {           // the projection is rounded first, then added (llama_glide.py:466)
                        const V4 r4 = *reinterpret_cast<const V4*>(p.residual + ((long)m * p.ldr + nn) * 2);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = E::from_f32(E::to_f32(o[e]) + E::to_f32(r4[e]));
                    }
                    *reinterpret_cast<V4*>(p.y + ((long)m * p.ldy + nn) * 2) = o;
                }
                // this tile's 16 columns of row m in the canonical order (every lane takes part in the shuffles)
                if (p.ssq_out != nullptr) tsq[mt] = ssq_tile16(ssq_quad(E::to_f32(o[0]), E::to_f32(o[1]), E::to_f32(o[2]), E::to_f32(o[3])));
            }
            if (p.ssq_out != nullptr) {                    // (uniform) the slab's 64 columns: tiles in wave order
                __syncthreads();                           // the 4-wave reduction no longer reads `red`
                if (g4 == 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) red[wave * (MT * 16) + mt * 16 + l15] = tsq[mt];
                }
                __syncthreads();
                if (tid < MT * 16 && tid < p.M) {
                    const int R = MT * 16;
                    p.ssq_out[(long)tid * (p.N >> 6) + slab * NPASS + h] = ssq_slab64(red[tid], red[R + tid], red[2 * R + tid], red[3 * R + tid]);
                }
                __syncthreads();
            }
        }
    }
}

// (round 4's persistent layer-tail launch lives in tools/mb/layer_tail_kernel.inc: measured not faster, diagnostic variants only)
#ifdef LS_WITH_TAIL
#define LS_TAIL_PART 1
#include "../../tools/mb/layer_tail_kernel.inc"
#undef LS_TAIL_PART
#endif

// ---- weight packing ---------------------------------------------------------------------
// The 64 rows of slab g at k-step s form one contiguous 4 KB block of 4 tiles:
// packed[((g * nks + s) * 4 + t) * 512 + l * 8 + e] = W[64 g + 16 t + (l & 15)][32 s + 8 (l >> 4) + e]
// (16-bit elements; rows >= N are zero).  A wave multiplying a slab reads 4 KB contiguous per k-step.
// `w_up` != null packs the gate/up pair of an MLP as ONE matrix of 2N rows whose 16-row tiles alternate: tile 2j =
// gate rows 16j.., tile 2j+1 = up rows 16j.. (the silu(gate)*up epilogue pairs neighbouring tiles).
// `rope` packs a q/k projection for EPI_QKV_ROPE: within every 128-row head the 8 tiles are stored in the order
// 0,4,1,5,2,6,3,7, so that rows d.. and 64+d.. (a rotary pair) are neighbouring tiles of one workgroup.
__global__ __launch_bounds__(256) void pack_weight_kernel(const uint16_t* __restrict__ w, const uint16_t* __restrict__ w_up,
                                                          uint16_t* __restrict__ out, int N, int K, long nblocks, int rope) {
    const int nks = K >> 5;
    for (long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblocks; blk += (long)gridDim.x * 4) {
        const int t = (int)(blk & 3);
        const long gs = blk >> 2;
        const int g = (int)(gs / nks), ks = (int)(gs % nks);
        const int l = threadIdx.x & 63;
        const int T = g * 4 + t;                           // 16-row tile of the packed matrix
        const uint16_t* src = w;
        int row = T * 16 + (l & 15);
        if (w_up != nullptr) {
            src = (T & 1) ? w_up : w;
            row = (T >> 1) * 16 + (l & 15);
        } else if (rope) {
            row = ((T & ~7) + ((T & 7) >> 1) + 4 * (T & 1)) * 16 + (l & 15);
        }
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < N) v = *reinterpret_cast<const uint4*>(src + (long)row * K + ks * 32 + (l >> 4) * 8);
        *reinterpret_cast<uint4*>(out + blk * 512 + (long)g * (GROUP_PAD / 2) + l * 8) = v;
    }
}

// ---- host side ---------------------------------------------------------------------------
int num_cus_gemm() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

struct Plan {
    int MT, NT, nslabs, S, nks, N, flag_off;
    size_t lds, counter_bytes, part_bytes;
};

int pick_mt(int M) { return M <= 16 ? 1 : M <= 32 ? 2 : M <= 80 ? 5 : 0; }

// The split count is a function of (N, K, epilogue) ONLY -- never of M -- so that a row's
// summation order is the same in every pass (see the header comment).  `groups` = 64-row groups of W.
int pick_splits(int groups, int nks, int forced) {
    const int smax = nks / 16 > 0 ? nks / 16 : 1;     // at least 512 k (8 chunks) per split
    if (forced > 0) return forced < smax ? forced : smax;
    const int cus = 256;
    if (groups >= cus) return 1;                      // a weight that already fills the chip is never split
    // Measured model of a launch (rocprofv3, M = 74, K = 4096 / 14336; fits within ~1 us):
    //   t = rounds x (bytes per workgroup / stream rate + 3 us of ramp) + 0.4 us x S of partial traffic,
    //   stream rate per workgroup = min(27 GB/s, 6 TB/s / resident workgroups), rounds = ceil(groups x S / 256).
    const double bytes_per_group = 64.0 * nks * 32 * 2;
    int best = 1;
    double best_t = 1e30;
    for (int S = 1; S <= smax && S <= 32; ++S) {
        const int wgs = groups * S;
        const int rounds = (wgs + cus - 1) / cus;
        const double rate = fmin(27e9, 6e12 / (wgs < cus ? wgs : cus));
        const double t = rounds * (bytes_per_group / S / rate + 3e-6) + 0.4e-6 * S;
        if (t < best_t - 1e-9) { best_t = t; best = S; }
    }
    return best;
}

int make_plan(const ls_linear_desc* d, Plan& pl) {
    if (!d || !d->x || !d->y || !d->w[0]) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: null pointer");
    if (d->dtype != LS_F16 && d->dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: dtype");
    if (d->M < 1 || d->K < 128 || d->K % 64 != 0)
        LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear: K=%d must be a multiple of 64, >= 128", d->K);
    pl.MT = pick_mt(d->M);
    if (pl.MT == 0) LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear: M=%d > 80 rows is a plain library GEMM, not this kernel", d->M);
    if (d->n_seg < 1 || d->n_seg > 3) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: n_seg");
    if (d->ldx < d->K) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: ldx < K");
    if ((d->ldx % 8) != 0 || (d->ldy % 4) != 0) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: ldx must be a multiple of 8, ldy of 4");
    int rows;                                           // rows of the packed matrix the grid covers
    if (d->epilogue == LS_EPI_SILU_MUL) {
        if (d->n_seg != 1) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: silu_mul takes ONE weight packed by ls_linear_pack_gate_up");
        if (d->n[0] < 1 || d->n[0] % 16 != 0) LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear: silu_mul needs N %% 16 == 0");
        pl.N = d->n[0];
        rows = 2 * d->n[0];
    } else if (d->epilogue == LS_EPI_NONE || d->epilogue == LS_EPI_QKV_ROPE) {
        if (d->epilogue == LS_EPI_QKV_ROPE) {
            if (!d->rope_cos || !d->rope_sin) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: qkv_rope needs the cos/sin tables");
            for (int i = 0; i < d->n_seg; ++i)
                if (d->n[i] % 128 != 0) LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear: qkv_rope segments are heads x 128 rows");
        }
        int N = 0;
        for (int i = 0; i < d->n_seg; ++i) {
            if (!d->w[i] || d->n[i] < 1) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: segment %d", i);
            if (d->n_seg > 1 && d->n[i] % 128 != 0) LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear: multi-segment rows must be multiples of 128");
            if (d->n[i] % 4 != 0) LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear: N %% 4");
            N += d->n[i];
        }
        pl.N = N;
        rows = N;
    } else {
        LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: epilogue");
    }
    if (d->ldy < pl.N) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: ldy < N");
    if (d->residual && (d->epilogue != LS_EPI_NONE || d->ldr < pl.N || (d->ldr % 4) != 0))
        LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: residual goes with LS_EPI_NONE, ldr >= N, ldr %% 4 == 0");
    if (d->norm_weight && (!d->ssq_in || d->ssq_parts < 1))
        LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: norm_weight needs the rows' sum-of-squares partials (ssq_in, ssq_parts)");
    if (d->ssq_out && (d->epilogue != LS_EPI_NONE || d->n_seg != 1 || pl.N % 64 != 0))
        LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear: ssq_out goes with LS_EPI_NONE, one segment, N %% 64 == 0");
    const int groups = (rows + 63) / 64;
    // more than 32 token rows run ONE workgroup per CU (512 registers per wave): give it 128 weight rows when that
    // still fills the chip -- half the x staging traffic, half the workgroups (measured: lm_head 230 -> 209 us)
    pl.NT = (pl.MT >= 5 && groups >= 2 * 224 && d->epilogue != LS_EPI_QKV_ROPE) ? 8 : 4;
    pl.nslabs = (rows + pl.NT * 16 - 1) / (pl.NT * 16);
    pl.nks = d->K / 32;
    pl.S = pick_splits(groups, pl.nks, d->n_splits);
    if (pl.NT == 8) pl.S = d->n_splits > 0 ? pl.S : 1;
    if ((size_t)pl.nslabs * 4 > COUNTER_BYTES) LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear: N too large (%d slabs)", pl.nslabs);
    pl.lds = (size_t)16 * 4 * pl.MT * 64 * 4 + 16;     // 4 waves x 4 tiles x MT accumulators (>= the 4 x-slabs) + the last-arriver flag
    pl.flag_off = (int)pl.lds - 16;
    if (d->norm_weight) pl.lds += (size_t)pl.MT * 64 + (size_t)d->K * 2;      // + 1 / rms of the rows + the norm weight
    if (pl.lds > 160 * 1024) LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear: K=%d too large for a folded norm", d->K);
    pl.counter_bytes = COUNTER_BYTES;
    pl.part_bytes = pl.S > 1 ? (size_t)pl.S * pl.nslabs * pl.NT * pl.MT * 4 * 64 * 4 : 0;
    return LS_OK;
}

template <typename E, int MT, int NT, int EPI, bool NORM>
int launch_n(const GemmK& k, const Plan& pl, hipStream_t s) {
    auto kern = skinny_gemm_kernel<E, MT, NT, EPI, NORM>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.nslabs, pl.S), dim3(GEMM_THREADS), pl.lds, s, k);
    LS_CHECK_LAUNCH("skinny_gemm_kernel");
    return LS_OK;
}

template <typename E, int MT, int NT, int EPI>
int launch(const GemmK& k, const Plan& pl, hipStream_t s) {
    return k.norm_w != nullptr ? launch_n<E, MT, NT, EPI, true>(k, pl, s) : launch_n<E, MT, NT, EPI, false>(k, pl, s);
}

template <typename E, int EPI>
int launch_mt(const GemmK& k, const Plan& pl, hipStream_t s) {
    switch (pl.MT) {
        case 1: return launch<E, 1, 4, EPI>(k, pl, s);
        case 2: return launch<E, 2, 4, EPI>(k, pl, s);
        default: return pl.NT == 8 ? launch<E, 5, 8, EPI>(k, pl, s) : launch<E, 5, 4, EPI>(k, pl, s);
    }
}

template <typename E, int EPI>
int launch_mt4(const GemmK& k, const Plan& pl, hipStream_t s) {      // epilogues that only exist with 4-tile slabs
    switch (pl.MT) {
        case 1: return launch<E, 1, 4, EPI>(k, pl, s);
        case 2: return launch<E, 2, 4, EPI>(k, pl, s);
        default: return launch<E, 5, 4, EPI>(k, pl, s);
    }
}

#ifdef LS_WITH_TAIL
#define LS_TAIL_PART 2
#include "../../tools/mb/layer_tail_kernel.inc"
#undef LS_TAIL_PART
#endif

}  // namespace

extern "C" {

size_t ls_linear_packed_bytes(int N, int K) {
    if (N < 1 || K < 32 || K % 32 != 0) return 0;
    return (size_t)((N + 63) / 64) * (64 * (size_t)K * 2 + (size_t)GROUP_PAD);
}

static int pack_impl(const void* w, const void* w_up, void* packed, int N, int K, int dtype, void* stream, const char* what,
                     int rope = 0) {
    if (!w || !packed) LS_FAIL(LS_ERR_INVALID_ARG, "%s: null pointer", what);
    if (dtype != LS_F16 && dtype != LS_BF16) LS_FAIL(LS_ERR_INVALID_ARG, "%s: dtype", what);
    if (N < 1 || K < 32 || K % 32 != 0) LS_FAIL(LS_ERR_UNSUPPORTED, "%s: K=%d must be a multiple of 32", what, K);
    const int rows = w_up ? 2 * N : N;
    const long nblocks = (long)((rows + 63) / 64) * 4 * (K / 32);
    long grid = (nblocks + 3) / 4;
    if (grid > 65535 * 4) grid = 65535 * 4;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t*>(w), static_cast<const uint16_t*>(w_up), static_cast<uint16_t*>(packed), N, K,
                       nblocks, rope);
    LS_CHECK_LAUNCH("pack_weight_kernel");
    return LS_OK;
}

int ls_linear_pack_weight(const void* weight, void* packed, int N, int K, int dtype, void* stream) {
    return pack_impl(weight, nullptr, packed, N, K, dtype, stream, "ls_linear_pack_weight");
}

int ls_linear_pack_gate_up(const void* gate_weight, const void* up_weight, void* packed, int N, int K, int dtype, void* stream) {
    if (!up_weight) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear_pack_gate_up: null pointer");
    if (N % 16 != 0) LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear_pack_gate_up: N %% 16");
    return pack_impl(gate_weight, up_weight, packed, N, K, dtype, stream, "ls_linear_pack_gate_up");
}

int ls_linear_pack_rope(const void* weight, void* packed, int N, int K, int dtype, void* stream) {
    if (N % 128 != 0) LS_FAIL(LS_ERR_UNSUPPORTED, "ls_linear_pack_rope: N must be heads x 128 rows");
    return pack_impl(weight, nullptr, packed, N, K, dtype, stream, "ls_linear_pack_rope", 1);
}

size_t ls_linear_workspace_bytes(const ls_linear_desc* d) {
    Plan pl;
    if (make_plan(d, pl) != LS_OK) return 0;
    return pl.counter_bytes + pl.part_bytes;
}

static int linear_launch(const ls_linear_desc* d, void* workspace, size_t workspace_bytes, void* stream, int prefetch_units) {
    Plan pl;
    int rc = make_plan(d, pl);
    if (rc != LS_OK) return rc;
    if (!workspace || workspace_bytes < pl.counter_bytes + pl.part_bytes)
        LS_FAIL(LS_ERR_WORKSPACE, "ls_linear_fwd: workspace %zu < %zu bytes", workspace_bytes, pl.counter_bytes + pl.part_bytes);
    GemmK k{};
    k.x = static_cast<const char*>(d->x);
    k.ldx = d->ldx;
    for (int i = 0; i < 3; ++i) {
        k.w[i] = i < d->n_seg ? static_cast<const char*>(d->w[i]) : nullptr;
        k.bias[i] = i < d->n_seg ? static_cast<const char*>(d->bias[i]) : nullptr;
        k.n[i] = i < d->n_seg ? d->n[i] : 0;
    }
    k.y = static_cast<char*>(d->y);
    k.ldy = d->ldy;
    k.counters = static_cast<unsigned*>(workspace);
    k.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + pl.counter_bytes);
    k.M = d->M;
    k.K = d->K;
    k.N = pl.N;
    k.nks = pl.nks;
    k.S = pl.S;
    k.nslabs = pl.nslabs;
    k.flag_off = pl.flag_off;
    k.rope_cos = static_cast<const char*>(d->rope_cos);
    k.rope_sin = static_cast<const char*>(d->rope_sin);
    k.rope_segs = d->n_seg < 2 ? d->n_seg : 2;
    k.residual = static_cast<const char*>(d->residual);
    k.ldr = d->ldr;
    k.norm_w = static_cast<const char*>(d->norm_weight);
    k.norm_eps = d->norm_eps;
    k.ssq_in = d->ssq_in;
    k.ssq_parts = d->ssq_parts;
    k.ssq_out = d->ssq_out;
    k.prefetch_units = prefetch_units;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d->ev_start) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_start), s);
    if (d->dtype == LS_F16)
        rc = d->epilogue == LS_EPI_SILU_MUL   ? launch_mt<ElemF16, EPI_SILU_MUL>(k, pl, s)
             : d->epilogue == LS_EPI_QKV_ROPE ? launch_mt4<ElemF16, EPI_QKV_ROPE>(k, pl, s)
                                              : launch_mt<ElemF16, EPI_NONE>(k, pl, s);
    else
        rc = d->epilogue == LS_EPI_SILU_MUL   ? launch_mt<ElemBF16, EPI_SILU_MUL>(k, pl, s)
             : d->epilogue == LS_EPI_QKV_ROPE ? launch_mt4<ElemBF16, EPI_QKV_ROPE>(k, pl, s)
                                              : launch_mt<ElemBF16, EPI_NONE>(k, pl, s);
    if (d->ev_stop) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_stop), s);
    return rc;
}

int ls_linear_fwd(const ls_linear_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    return linear_launch(d, workspace, workspace_bytes, stream, 0);
}

int ls_linear_prefetch(const ls_linear_desc* d, int units, void* workspace, size_t workspace_bytes, void* stream) {
    if (units < 1) LS_FAIL(LS_ERR_INVALID_ARG, "ls_linear_prefetch: units");
    return linear_launch(d, workspace, workspace_bytes, stream, units);
}

#ifdef LS_WITH_TAIL
#define LS_TAIL_PART 3
#include "../../tools/mb/layer_tail_kernel.inc"
#undef LS_TAIL_PART
#endif

}  // extern "C"
