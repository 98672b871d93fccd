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
    static __device__ __forceinline__ T from_f32(float x) { return (T)x; }
    static __device__ __forceinline__ float to_f32(T x) { return (float)x; }
};
struct ElemBF16 {
    using T = __bf16;
    using V8 = bf16x8;
    using V4 = bf16x4;
    static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ T from_f32(float x) { return (T)x; }
    static __device__ __forceinline__ float to_f32(T x) { return (float)x; }
};

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
