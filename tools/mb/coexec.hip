#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode bit0: waves 0-3 do MFMA; bit1: waves 4-7 do VALU (exp+fma); 8 waves per WG = 2 per SIMD
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int same_wave) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    const bool do_mfma = same_wave ? (mode & 1) : ((mode & 1) && wave < 4);
    const bool do_valu = same_wave ? (mode & 2) : ((mode & 2) && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
                v[i] = __builtin_amdgcn_exp2f(v[i] * 0.01f) + v[i];
                v[i] = __builtin_fmaf(v[i], 0.5f, 0.25f);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int same = 0; same < 2; ++same)
        for (int mode = 1; mode <= 3; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 1000, mode, same);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 20000, mode, same);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("same_wave=%d mode=%d (1=MFMA only, 2=VALU only, 3=both): %.3f ms\n", same, mode, ms);
        }
    return 0;
}
