#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode 0: VALU only, 1: MFMA only, 2: waves alternate (even waves VALU, odd waves MFMA) on the same SIMDs
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool doValu = mode == 0 || (mode == 2 && (wave & 1) == 0);
    const bool doMfma = mode == 1 || (mode == 2 && (wave & 1) == 1);
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c0 = 0.1f, c1 = 0.2f, c2 = 0.3f, c3 = 0.4f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    if (doValu) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { c0 = __builtin_fmaf(c0, b, a); c1 = __builtin_fmaf(c1, b, a); c2 = __builtin_fmaf(c2, b, a); c3 = __builtin_fmaf(c3, b, a); }
        }
    }
    if (doMfma) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + acc0[0] + acc1[1];
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode);   // 8 waves per CU = 2 per SIMD
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.3f ms  (per wave: %s)\n", mode, ms, mode == 0 ? "32 FMA x iters" : mode == 1 ? "4 MFMA x iters" : "half the waves each");
        }
    }
    return 0;
}
