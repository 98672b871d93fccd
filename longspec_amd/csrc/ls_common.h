// Shared device/host helpers for liblongspec_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/longspec_hip.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LS_HEAD_DIM 128
#define LS_WAVE 64

// ---- error plumbing (host) ---------------------------------------------------
void ls_set_error(const char* fmt, ...);
#define LS_FAIL(code, ...)          \
    do {                            \
        ls_set_error(__VA_ARGS__);  \
        return (code);              \
    } while (0)
#define LS_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) LS_FAIL(LS_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---- element traits ------------------------------------------------------------
struct ElemF16 {
    using T = _Float16;
    using V8 = f16x8;
    using V4 = f16x4;
    static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    // Register-class-pinned forms for kernels that own the whole 512-entry file (one wave per SIMD): hipcc does not split a
    // wave's values over the two halves by itself (it spilled 2.8 KB per lane rather than use the accumulator half).
    //   mfma_acc_a: C/D tied in the ACCUMULATOR half ("a"), A and B in the vector half
    //   mfma_b_a:   B operand read from the accumulator half, C/D in the vector half
    // (an MFMA result feeding the next MFMA as C needs no wait state; every other reader is hundreds of instructions away.  The
    // `s_nop 1` in front: the compiler may have WRITTEN an operand just before the statement -- a copy that assembles a 128-bit
    // fragment, the zeroing of an accumulator -- and does not know that what follows is an MFMA that needs two wait states
    // behind a vector write of its operands; without it the results were wrong in a timing-dependent way)
#if !defined(LS_W4_ASM_MFMA) || defined(LS_W4_BUILTIN_PV)
    static __device__ __forceinline__ void mfma_acc_a(V8 a, V8 b, f32x4& c) { c = mfma(a, b, c); }
#else
    static __device__ __forceinline__ void mfma_acc_a(V8 a, V8 b, f32x4& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
#endif
#if !defined(LS_W4_ASM_MFMA) || defined(LS_W4_BUILTIN_QK)
    static __device__ __forceinline__ void mfma_b_a(V8 a, V8 b, f32x4& c) { c = mfma(a, b, c); }
#else
    static __device__ __forceinline__ void mfma_b_a(V8 a, V8 b, f32x4& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
    }
#endif
    static __device__ __forceinline__ T from_f32(float x) { return (T)x; }
    static __device__ __forceinline__ float to_f32(T x) { return (float)x; }
    // c + the sum of the 8 elements (fp32 accumulation): four v_dot2_f32_f16 against (1, 1)
    static __device__ __forceinline__ float sum8(V8 v, float c) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_fdot2(h2{v[2 * i], v[2 * i + 1]}, one, c, false);
        return c;
    }
};
struct ElemBF16 {
    using T = __bf16;
    using V8 = bf16x8;
    using V4 = bf16x4;
    static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
#if !defined(LS_W4_ASM_MFMA) || defined(LS_W4_BUILTIN_PV)
    static __device__ __forceinline__ void mfma_acc_a(V8 a, V8 b, f32x4& c) { c = mfma(a, b, c); }
#else
    static __device__ __forceinline__ void mfma_acc_a(V8 a, V8 b, f32x4& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
#endif
#if !defined(LS_W4_ASM_MFMA) || defined(LS_W4_BUILTIN_QK)
    static __device__ __forceinline__ void mfma_b_a(V8 a, V8 b, f32x4& c) { c = mfma(a, b, c); }
#else
    static __device__ __forceinline__ void mfma_b_a(V8 a, V8 b, f32x4& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
    }
#endif
    static __device__ __forceinline__ T from_f32(float x) { return (T)x; }
    static __device__ __forceinline__ float to_f32(T x) { return (float)x; }
    static __device__ __forceinline__ float sum8(V8 v, float c) {      // (bf16 -> fp32 is a shift: pairs, then the chain)
#pragma unroll
        for (int i = 0; i < 4; ++i) c += (float)v[2 * i] + (float)v[2 * i + 1];
        return c;
    }
};

// ---- canonical order of a row's sum of squares (RMSNorm), shared by misc.hip::rmsnorm_rows_kernel and the producers /
// consumers of gemm.hip so that a norm folded into the next projection is bit-identical to the stand-alone kernel:
//   quad  = ((x0^2 + x1^2) + x2^2) + x3^2                 4 consecutive columns (fp32, no contraction)
//   tile  = (quad0 + quad1) + (quad2 + quad3)             16 columns
//   slab  = ((tile0 + tile1) + tile2) + tile3             64 columns
//   group = ((((slab0 + slab1) + slab2) + ...) + slab7)    8 consecutive slabs = 512 columns (the last group may be short)
//   row   = (((group0 + group1) + group2) + ...)           in column order
__device__ __forceinline__ float ssq_quad(float a, float b, float c, float d) { return ((a * a + b * b) + c * c) + d * d; }
// MFMA accumulator layout: the quads of a row's 16 columns sit in lanes l, l+16, l+32, l+48
__device__ __forceinline__ float ssq_tile16(float quad) {
    const float pair = quad + __shfl_xor(quad, 16);        // (q0 + q1) in lanes of g4 = 0, 1; (q2 + q3) in g4 = 2, 3
    return pair + __shfl_xor(pair, 32);                    // commutative: the same bits in all four lanes
}
__device__ __forceinline__ float ssq_slab64(float t0, float t1, float t2, float t3) { return ((t0 + t1) + t2) + t3; }
// `slabs[0 .. n)`: a row's slab sums in column order (global or LDS); eight independent chains, then one
__device__ __forceinline__ float ssq_row(const float* slabs, int n) {
    float tot = 0.f;
    int b = 0;
    for (; b + 64 <= n; b += 64) {                          // 8 whole groups at a time
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            g[j] = slabs[b + 8 * j];
#pragma unroll
            for (int i = 1; i < 8; ++i) g[j] += slabs[b + 8 * j + i];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) tot += g[j];
    }
    for (; b < n; b += 8) {
        float g = slabs[b];
        for (int i = 1; i < 8 && b + i < n; ++i) g += slabs[b + i];
        tot += g;
    }
    return tot;
}

// round-trip through the storage type (one rounding), as `tensor.to(dtype)` does
template <typename E>
__device__ __forceinline__ float round_to(float x) {
    return E::to_f32(E::from_f32(x));
}

__device__ __forceinline__ float wave_xor_max_16_32(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ float wave_xor_sum_16_32(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// ---- peer exchange (xgmi.hip; attn.hip fuses it into the reduce / finish kernels of a sharded attention call) ----
constexpr int XCHG_MAX_WORLD = LS_XCHG_MAX_WORLD;
constexpr size_t XCHG_DATA_OFF = 256;             // flags: u64 [2][16] = 256 bytes, then f32 [2][world][cap_floats]
constexpr unsigned XCHG_SPIN_LIMIT = 1u << 20;    // default, x s_sleep(32) ~ 1 s (rank skew is milliseconds): a dead peer must not hang the GPU

struct XCtl {
    unsigned long long epoch;                      // the NEXT exchange to push; counted on the device (graph replays advance it):
                                                   // a push reads e and leaves e + 1, the wait behind it works on epoch - 1
    unsigned int error;                            // latched: a wait gave up
    unsigned int spin_limit;                       // polls (x s_sleep(32)) before a wait gives up
    unsigned int arrive_all;
    unsigned int arrive_push[XCHG_MAX_WORLD];
};

struct ls_xchg {
    int rank, world;
    size_t cap_floats, box_bytes;
    char* box;                                     // local mailbox
    XCtl* ctl;
    char** peers_dev;                              // device array [world] of mailbox addresses as this process maps them
    char* peers[XCHG_MAX_WORLD];
    bool connected;
};

// Stores into a mailbox (local or a peer's): write-through at system scope (sc0 sc1), so that "all my stores are
// acknowledged" is s_waitcnt vmcnt(0) and not a release fence -- a system-scope release writes the whole L2 back
// (the split partials of the attention call are still dirty in it: measured +35 us per call with fences).
__device__ __forceinline__ void xchg_store16(void* p, f32x4 v) {
    // (s_nop 1: a VALU write of a > 64-bit store's data registers needs two wait states behind the store -- the compiler
    // inserts them for its own stores only, and is free to reuse `v`'s registers right behind this statement)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void xchg_store4(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void xchg_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void xchg_raise_flag(char* box, int parity, int rank, unsigned long long epoch) {
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(box) + parity * XCHG_MAX_WORLD + rank;
    __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (ordered behind the data by the waits above)
}

// Loads of a record that a PEER stored into the local mailbox: SYSTEM scope (sc0 sc1 = aux 17 of the raw buffer load).  `nt` is
// only a streaming hint on gfx94x / gfx950; what keeps a stale line of this CU's L1 (or a non-coherent L2 line) out of the
// merge is the scope of the load.  Through the compiler's builtin, so that its s_waitcnt bookkeeping covers them.
// `bytes` < 2 GB (a mailbox slot is a few MB).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xchg_record_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes < 0x7fffffffull ? bytes : 0x7fffffffull), 0x00020000);
}
__device__ __forceinline__ f32x4 xchg_load16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 17));
}
__device__ __forceinline__ float xchg_load4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 17));
}

// poll flag[parity][src] of the local mailbox until rank `src` has delivered epoch `epoch` (one thread per source)
__device__ __forceinline__ void xchg_wait_flag(const char* box, XCtl* ctl, int parity, int src, unsigned long long epoch) {
    const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(box) + parity * XCHG_MAX_WORLD + src;
    unsigned spins = 0;
    if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;     // latched: never wait twice
    // RELAXED polls and no acquire behind them: the mailbox is UNCACHED device memory (ls_xchg_create), the merge reads the
    // records with system-scope loads (xchg_load16 / xchg_load4: served beyond this CU's L1 whatever it holds -- `nt`, used in
    // round 5, is only a cache-policy hint), those loads are issued behind this loop and behind the caller's barrier, and
    // everything else the merge reads was written by earlier kernels of this stream.  Validated on gfx950 (MI355X) with two
    // ranks on ONE GPU only -- no multi-GPU node was available in rounds 1-6.  An acquiring poll is `buffer_inv sc0 sc1` per poll
    // in every workgroup -- it throws the tree part, written by the launch in front, out of the L2 (round 4: 23.7 -> 22.0 us per
    // exchange at 16k rows per rank, profiles/r4_xchg_arrive_relaxed.json).
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > ctl->spin_limit) {
            __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
