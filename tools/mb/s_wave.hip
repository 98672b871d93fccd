// How should the S wave of the warp-specialised attention kernel lay out one step (40 QK^T MFMAs + the soft-max
// VALU work of 40 scores per lane: 40 fma, 40 exp2, ~80 simple ops) when the SIMD's other wave (the O wave) issues
// 40 MFMAs per step of its own?  All instruction order is pinned with asm volatile.
//   S patterns: 0 = 40 MFMAs, then the VALU block      1 = (1 MFMA, 1 fma, 1 exp, 2 add) x 40
//               2 = VALU only                           3 = MFMAs only
//               4 = (2 MFMAs, 2 fma, 2 exp, 4 add) x 20 5 = (4 MFMAs, then 16 VALU) x 10
//               6 / 7 = patterns 0 / 1 with the CURRENT mix of the kernel (row sums on the matrix pipe): 40 fma, 40 exp,
//                       20 packed conversions, no adds; the O wave then issues 45 MFMAs per step (O pattern 3)
//   O patterns: 0 = absent (4 waves per workgroup)      1 = 40 MFMAs per step        2 = idle wave
//   hipcc --offload-arch=gfx950 -O3 tools/mb/s_wave.hip -o s_wave && ./s_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f16x8 rnd(unsigned s) {
    f16x8 r;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        r[i] = (_Float16)(((int)(s >> 8) % 2000 - 1000) * 0.001f);
    }
    return r;
}

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(d, x, c, m) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(c), "v"(m))
#define EXP(d, x) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(x))
#define ADD(d, x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(x))
#define CVT(d, x, y) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))

template <int SP, int OP>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f16x8 a[4], b[5];
    for (int i = 0; i < 4; ++i) a[i] = rnd(threadIdx.x * 7 + i);
    for (int i = 0; i < 5; ++i) b[i] = rnd(threadIdx.x * 13 + i + 99);
    f32x4 acc[10];
    for (int i = 0; i < 10; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float s = 0;
    if (wave < 4) {
        float x[40], t[40], e[40], sum0 = 0.f, sum1 = 0.f;
        const float c = 0.001f * threadIdx.x, m = -0.5f;
        for (int i = 0; i < 40; ++i) x[i] = 0.01f * i + 0.001f * threadIdx.x;
        for (int it = 0; it < iters; ++it) {
            if constexpr (SP == 0) {
#pragma unroll
                for (int i = 0; i < 40; ++i) MFMA(acc[i % 10], a[i / 10], b[i % 5]);
#pragma unroll
                for (int i = 0; i < 40; ++i) { FMA(t[i], x[i], c, m); EXP(e[i], t[i]); ADD(sum0, e[i]); ADD(sum1, e[i]); }
            } else if constexpr (SP == 1) {
#pragma unroll
                for (int i = 0; i < 40; ++i) { MFMA(acc[i % 10], a[i / 10], b[i % 5]); FMA(t[i], x[i], c, m); EXP(e[i], t[i]); ADD(sum0, e[i]); ADD(sum1, e[i]); }
            } else if constexpr (SP == 2) {
#pragma unroll
                for (int i = 0; i < 40; ++i) { FMA(t[i], x[i], c, m); EXP(e[i], t[i]); ADD(sum0, e[i]); ADD(sum1, e[i]); }
            } else if constexpr (SP == 3) {
#pragma unroll
                for (int i = 0; i < 40; ++i) MFMA(acc[i % 10], a[i / 10], b[i % 5]);
            } else if constexpr (SP == 4) {
#pragma unroll
                for (int i = 0; i < 40; i += 2) {
                    MFMA(acc[i % 10], a[i / 10], b[i % 5]); MFMA(acc[(i + 1) % 10], a[(i + 1) / 10], b[(i + 1) % 5]);
                    FMA(t[i], x[i], c, m); FMA(t[i + 1], x[i + 1], c, m); EXP(e[i], t[i]); EXP(e[i + 1], t[i + 1]);
                    ADD(sum0, e[i]); ADD(sum1, e[i + 1]); ADD(sum0, e[i]); ADD(sum1, e[i + 1]);
                }
            } else if constexpr (SP == 6) {
                unsigned pk[20];
#pragma unroll
                for (int i = 0; i < 40; ++i) MFMA(acc[i % 10], a[i / 10], b[i % 5]);
#pragma unroll
                for (int i = 0; i < 40; ++i) { FMA(t[i], x[i], c, m); EXP(e[i], t[i]); if (i & 1) { CVT(pk[i / 2], e[i - 1], e[i]); } }
                sum0 += __uint_as_float(pk[it % 20]) * 0.f;
            } else if constexpr (SP == 7) {
                unsigned pk[20];
#pragma unroll
                for (int i = 0; i < 40; ++i) { MFMA(acc[i % 10], a[i / 10], b[i % 5]); FMA(t[i], x[i], c, m); EXP(e[i], t[i]); if (i & 1) { CVT(pk[i / 2], e[i - 1], e[i]); } }
                sum0 += __uint_as_float(pk[it % 20]) * 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < 40; i += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) MFMA(acc[(i + u) % 10], a[(i + u) / 10], b[(i + u) % 5]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { FMA(t[i + u], x[i + u], c, m); EXP(e[i + u], t[i + u]); ADD(sum0, e[i + u]); ADD(sum1, e[i + u]); }
                }
            }
        }
        s = sum0 + sum1;
    } else if constexpr (OP == 1 || OP == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < (OP == 3 ? 45 : 40); ++i) MFMA(acc[i % 10], a[(i / 10) & 3], b[i % 5]);
        }
    }
    for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SP, int OP>
void run(const char* what) {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000, threads = OP == 0 ? 256 : 512;
    hipLaunchKernelGGL((k<SP, OP>), dim3(256), dim3(threads), 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SP, OP>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-62s %8.1f ns/step\n", what, ms * 1e6 / iters);
    hipFree(out);
}

int main() {
    run<3, 0>("S: 40 MFMA                      O: absent");
    run<2, 0>("S: VALU (40 fma, 40 exp, 80 add)  O: absent");
    run<0, 0>("S: 40 MFMA then VALU            O: absent");
    run<1, 0>("S: (1 MFMA + 4 VALU) x 40       O: absent");
    run<4, 0>("S: (2 MFMA + 8 VALU) x 20       O: absent");
    run<5, 0>("S: (4 MFMA + 16 VALU) x 10      O: absent");
    run<3, 1>("S: 40 MFMA                      O: 40 MFMA");
    run<2, 1>("S: VALU                         O: 40 MFMA");
    run<0, 1>("S: 40 MFMA then VALU            O: 40 MFMA");
    run<1, 1>("S: (1 MFMA + 4 VALU) x 40       O: 40 MFMA");
    run<4, 1>("S: (2 MFMA + 8 VALU) x 20       O: 40 MFMA");
    run<5, 1>("S: (4 MFMA + 16 VALU) x 10      O: 40 MFMA");
    run<6, 0>("S: 40 MFMA then VALU (100: no adds) O: absent");
    run<6, 3>("S: 40 MFMA then VALU (100: no adds) O: 45 MFMA");
    run<7, 3>("S: (1 MFMA + 2.5 VALU) x 40      O: 45 MFMA");
    return 0;
}
