// Rate of v_pk_fma_f32 (two f32 FMAs per lane per instruction) on gfx950. 256 blocks x 1024 threads; mode 0: 32 scalar FMAs per
// iteration in 4 independent chains — which hipcc's SLP vectoriser turns into 16 v_pk_fma_f32 by itself —, mode 1: 32 v_pk_fma_f32.
// Measured on MI355X: mode 0 3.23 ms (104 TFLOP/s), mode 1 5.57 ms (121 TFLOP/s = 77 % of the 157 TFLOP/s packed peak; ~5 cycles
// per v_pk_fma_f32 and SIMD). Used to price a packed-f32 table build in k_superpose_mfma (rejected: registers).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/r02_microbench_packed_f32.hip -o /tmp/pk && /tmp/pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k(float* out, int iters, int mode) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float c0 = 0.1f, c1 = 0.2f, c2 = 0.3f, c3 = 0.4f;
    f2 p0 = {0.1f, 0.5f}, p1 = {0.2f, 0.6f}, p2 = {0.3f, 0.7f}, p3 = {0.4f, 0.8f};
    const f2 pa = {a, a + 1.0f}, pb = {b, b};
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { c0 = __builtin_fmaf(c0, b, a); c1 = __builtin_fmaf(c1, b, a); c2 = __builtin_fmaf(c2, b, a); c3 = __builtin_fmaf(c3, b, a); }
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p0 = __builtin_elementwise_fma(p0, pb, pa); p1 = __builtin_elementwise_fma(p1, pb, pa);
                p2 = __builtin_elementwise_fma(p2, pb, pa); p3 = __builtin_elementwise_fma(p3, pb, pa);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, iters, mode);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double flops = 2.0 * (mode ? 2 : 1) * 32.0 * iters * 256 * 1024;
        printf("mode %d (%s): %.3f ms, %.1f TFLOP/s\n", mode, mode ? "v_pk_fma_f32" : "v_fma_f32", ms, flops / ms / 1e9);
    }
    return 0;
}
