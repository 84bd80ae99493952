// Do vector FMAs and f32 MFMAs of DIFFERENT waves overlap on one SIMD of gfx950? (round 1's version alternated the kinds by wave
// parity; with waves dealt round-robin to the 4 SIMDs of a CU that puts two waves of the SAME kind on every SIMD, so it measured
// nothing. Here the kind is chosen by (wave >> shift) & 1 for shift = 0, 1, 2 — one of them co-locates the kinds whatever the dealing —
// and every wave reports the SIMD it ran on (HW_ID) so the placement is known, not assumed.)
// hipcc --offload-arch=gfx950 -O3 profiles/r02_microbench_valu_mfma_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// kind of a wave: mode 0 all VALU, 1 all MFMA, 2 + s: (wave >> s) & 1 ? MFMA : VALU
__global__ __launch_bounds__(512) void k(float* out, unsigned* where, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool mf = mode == 1 || (mode >= 2 && ((wave >> (mode - 2)) & 1));
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c0 = 0.1f, c1 = 0.2f, c2 = 0.3f, c3 = 0.4f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    if (!mf) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { c0 = __builtin_fmaf(c0, b, a); c1 = __builtin_fmaf(c1, b, a); c2 = __builtin_fmaf(c2, b, a); c3 = __builtin_fmaf(c3, b, a); }
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + acc0[0] + acc1[1];
    if ((threadIdx.x & 63) == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        where[blockIdx.x * 8 + wave] = (hw & 0xFFFFu) | ((unsigned)mf << 31);    // wave_id [3:0], simd_id [5:4], ..., cu_id [11:8]
    }
}
int main() {
    float* d; unsigned* w;
    hipMalloc(&d, 256 * 512 * 4); hipMalloc(&w, 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 5; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, w, iters, mode);   // 8 waves per CU = 2 per SIMD
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned> h(256 * 8);
        hipMemcpy(h.data(), w, h.size() * 4, hipMemcpyDeviceToHost);
        // per block: how many SIMDs hold one wave of each kind
        int mixed = 0, total = 0;
        for (int b = 0; b < 256; ++b) {
            int nV[4] = {0, 0, 0, 0}, nM[4] = {0, 0, 0, 0};
            for (int v = 0; v < 8; ++v) { const unsigned x = h[b * 8 + v]; const int simd = (x >> 4) & 3; if (x >> 31) ++nM[simd]; else ++nV[simd]; }
            for (int s = 0; s < 4; ++s) { total += (nV[s] + nM[s]) > 0; mixed += nV[s] > 0 && nM[s] > 0; }
        }
        printf("mode %d: %.3f ms   SIMDs holding both kinds: %d of %d   (block 0 wave->simd:", mode, ms, mixed, total);
        for (int v = 0; v < 8; ++v) printf(" %u%c", (h[v] >> 4) & 3, (h[v] >> 31) ? 'M' : 'V');
        printf(")\n");
    }
    return 0;
}
