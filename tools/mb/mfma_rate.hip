// Issue rate of the two fp16 MFMA shapes on gfx950 with NON-ZERO random operands, 1 and 2 waves per SIMD.
// (The attention kernels are matrix-pipe bound: this sets what "bound" means for v_mfma_f32_16x16x32_f16 vs
// v_mfma_f32_32x32x16_f16.)     hipcc --offload-arch=gfx950 -O3 tools/mb/mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline f16x8 rnd(unsigned s) {
    f16x8 r;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        r[i] = (_Float16)(((int)(s >> 8) % 2000 - 1000) * 0.001f);
    }
    return r;
}

template <int SHAPE, int NACC>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(threadIdx.x * 7 + i); b[i] = rnd(threadIdx.x * 13 + i + 99); }
    float s = 0;
    unsigned long long t0, t1;
    if constexpr (SHAPE == 16) {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, int NACC>
void run(int threads, const char* what) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, cyc, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC;                 // MFMAs per wave
    const double waves_per_simd = threads / 256.0;
    const double flop = (SHAPE == 16 ? 16384.0 : 32768.0);
    const double tf = n * (threads / 64) * 256 * flop / (ms * 1e-3) / 1e12;
    printf("%-34s %d thr: %.3f ms  %.1f ticks/MFMA/wave  %.1f ticks per MFMA per SIMD  %.0f TFLOP/s  (%.2f GHz if ticks are shader cycles)\n",
           what, threads, ms, c / n, c / n / waves_per_simd, tf, c / (ms * 1e-3) / 1e9);
}

int main() {
    run<16, 16>(256, "16x16x32 f16, 1 wave/SIMD");
    run<16, 16>(512, "16x16x32 f16, 2 waves/SIMD");
    run<32, 4>(256, "32x32x16 f16, 1 wave/SIMD");
    run<32, 4>(512, "32x32x16 f16, 2 waves/SIMD");
    run<32, 8>(256, "32x32x16 f16, 1 wave/SIMD, 8 acc");
    return 0;
}
