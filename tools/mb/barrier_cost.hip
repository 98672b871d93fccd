// Price of s_barrier in an 8-wave workgroup (one per CU) on gfx950: a bare loop of barriers, and barriers between MFMA bursts.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NM, int NB>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.1f); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 15]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < NB; ++i) asm volatile("s_barrier" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NM, int NB>
void run(int threads) {
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<NM, NB>), dim3(256), dim3(threads), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NB>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%d waves: %2d MFMA + %d s_barrier per iteration: %.1f ns per iteration\n", threads / 64, NM, NB, ms * 1e6 / iters);
}
int main() {
    run<0, 1>(512); run<0, 2>(512); run<16, 0>(512); run<16, 1>(512); run<16, 2>(512); run<32, 1>(512); run<32, 2>(512);
    run<0, 1>(256); run<16, 0>(256); run<16, 1>(256);
    return 0;
}
