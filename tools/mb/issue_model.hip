// How many VALU instructions hide behind one MFMA on gfx950?  Round 3: the ping-pong attention kernel measured its two waves
// of a SIMD (one in a pure-MFMA segment, one in a pure-VALU segment) at the SUM of their stand-alone times with
// v_mfma_f32_16x16x32_f16.  This benchmark prices, for both fp16 MFMA shapes:
//   A. one wave per SIMD issuing {1 MFMA, k independent VALU fillers} (k = 0..10; fillers = v_fma_f32 or v_exp_f32)
//   B. two waves per SIMD: wave w = pure MFMA stream, wave w+4 = pure VALU stream (n VALU per MFMA of the partner)
// Output: ns per MFMA per SIMD.      hipcc --offload-arch=gfx950 -O3 tools/mb/issue_model.hip -o issue_model && ./issue_model
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline f16x8 rnd(unsigned s) {
    f16x8 r;
    for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; r[i] = (_Float16)(((int)(s >> 8) % 2000 - 1000) * 0.001f); }
    return r;
}

// SHAPE 16: 16x16x32 (16 accumulators of 4), SHAPE 32: 32x32x16 (4 accumulators of 16).  K fillers per MFMA, EXP of them v_exp_f32.
// ROLE_SPLIT = 0: every wave runs the mixed stream.  1: waves 0-3 MFMA only, waves 4-7 the fillers only (K per partner MFMA).
template <int SHAPE, int K, int EXPS, int ROLE_SPLIT>
__global__ __launch_bounds__(512) void kern(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(threadIdx.x * 7 + i); b[i] = rnd(threadIdx.x * 13 + i + 99); }
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 0.001f + i * 0.01f;
    const bool do_m = ROLE_SPLIT ? wave < 4 : true, do_v = ROLE_SPLIT ? wave >= 4 : true;
    float s = 0;
    constexpr int NM = SHAPE == 16 ? 16 : 8;          // MFMAs per loop body
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (do_m) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
                if (do_v) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if (k < EXPS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i * K + k) % 12]));
                        else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i * K + k) % 12]) : "v"(v[11]));
                    }
                }
            }
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (do_m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
                if (do_v) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if (k < EXPS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i * K + k) % 12]));
                        else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i * K + k) % 12]) : "v"(v[11]));
                    }
                }
            }
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    }
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// B (clean): the role is decided ONCE; waves 0-3 run a bare MFMA loop (16 MFMAs of 16x16x32 or 8 of 32x32x16 per iteration), waves 4-7 a bare
// filler loop (NF fillers per iteration, EXPS of every 4 are v_exp_f32).  Both run `iters` iterations.
template <int SHAPE, int NF, int EXPS>
__global__ __launch_bounds__(512) void kern_split(float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0;
    if (wave < 4) {
        f16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = rnd(threadIdx.x * 7 + i); b[i] = rnd(threadIdx.x * 13 + i + 99); }
        if constexpr (SHAPE == 16) {
            f32x4 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
            }
            for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        } else {
            f32x16 acc[8];
            for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
            }
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
        }
    } else {
        float v[12];
        for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 0.001f + i * 0.01f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                if ((k & 3) < EXPS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k % 12]));
                else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k % 12]) : "v"(v[11]));
            }
        }
        for (int i = 0; i < 12; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NF, int EXPS>
void run_split() {
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 3000;
    hipLaunchKernelGGL((kern_split<SHAPE, NF, EXPS>), dim3(256), dim3(512), 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern_split<SHAPE, NF, EXPS>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s split roles: partner issues %d fillers (%d of 4 exp) per 512 kflop of MFMA: %.1f ns per iteration (MFMA stream alone: see NF=0)\n",
           SHAPE == 16 ? "16x16x32" : "32x32x16", NF, EXPS, ms * 1e6 / iters);
}

template <int SHAPE, int K, int EXPS, int ROLE_SPLIT>
void run(int threads) {
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 3000;
    hipLaunchKernelGGL((kern<SHAPE, K, EXPS, ROLE_SPLIT>), dim3(256), dim3(threads), 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<SHAPE, K, EXPS, ROLE_SPLIT>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nm = SHAPE == 16 ? 16 : 8;
    const double mf_per_simd = (double)iters * nm * (ROLE_SPLIT ? 1 : threads / 256);
    const double ns = ms * 1e6 / mf_per_simd;
    const double flop = SHAPE == 16 ? 16384.0 : 32768.0;
    printf("%s %s waves/SIMD=%d fillers/MFMA=%d (exp %d): %.2f ns per MFMA per SIMD = %.2f ns per 32-kflop, %.0f TFLOP/s\n",
           SHAPE == 16 ? "16x16x32" : "32x32x16", ROLE_SPLIT ? "split-roles" : "mixed-stream", threads / 256, K, EXPS, ns, ns * 32768.0 / flop,
           flop * 1024 / ns / 1e3);
}

int main() {
    printf("--- A. one wave per SIMD, mixed stream, fillers = v_fma_f32\n");
    run<16, 0, 0, 0>(256); run<16, 1, 0, 0>(256); run<16, 2, 0, 0>(256); run<16, 3, 0, 0>(256); run<16, 4, 0, 0>(256); run<16, 6, 0, 0>(256);
    run<32, 0, 0, 0>(256); run<32, 2, 0, 0>(256); run<32, 4, 0, 0>(256); run<32, 5, 0, 0>(256); run<32, 6, 0, 0>(256); run<32, 8, 0, 0>(256); run<32, 10, 0, 0>(256);
    printf("--- A'. same, half of the fillers v_exp_f32\n");
    run<16, 2, 1, 0>(256); run<16, 4, 2, 0>(256); run<32, 4, 2, 0>(256); run<32, 6, 3, 0>(256); run<32, 8, 4, 0>(256);
    printf("--- A''. two waves per SIMD, both the mixed stream\n");
    run<16, 0, 0, 0>(512); run<16, 2, 0, 0>(512); run<16, 2, 1, 0>(512); run<32, 0, 0, 0>(512); run<32, 4, 0, 0>(512); run<32, 4, 2, 0>(512); run<32, 6, 3, 0>(512);
    printf("--- B. two waves per SIMD, split roles: waves 0-3 a bare MFMA stream (262 kflop per iteration), waves 4-7 a bare filler stream\n");
    run_split<16, 0, 0>(); run_split<16, 8, 0>(); run_split<16, 16, 0>(); run_split<16, 32, 0>(); run_split<16, 48, 0>(); run_split<16, 64, 0>();
    run_split<16, 16, 2>(); run_split<16, 32, 2>(); run_split<16, 48, 2>();
    run_split<32, 0, 0>(); run_split<32, 16, 0>(); run_split<32, 32, 0>(); run_split<32, 48, 0>(); run_split<32, 64, 0>();
    run_split<32, 16, 2>(); run_split<32, 32, 2>(); run_split<32, 48, 2>();
    return 0;
}
