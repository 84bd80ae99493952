// K7u: the superposition of a field whose every (layer, step) slice has ONE sigma over its live rays — a water phantom, the reference's
// own WATER_CUBE_TEST. There the per-voxel-sigma patches of kernelSuperposition (kernel_wrapper.cuh:432-489) add up to a separable
// convolution of the slice with that layer's pixel-integrated Gaussian: 2 (2 rho + 1) multiply-adds per pixel instead of (2 rho + 1)^2 —
// what the reference's CPU path does (cpu_convolution_1d.cpp: xConvCpuScat + yConvCpu), and the HIP result is checked against exactly
// that code (tests/test_gpu_parity.py, C1). Whether a field qualifies is decided on the device (k_fill raises FieldState::nonUniform when
// the live rays of a tile differ in sigma^2, k_ks_plan sets uniformField when no tile did and every depositing slice has one sigma^2 over
// all its tiles); these launches return at once otherwise, the general superposition when it does.
//
// Both passes are banded Toeplitz products on the matrix cores (v_mfma_f32_16x16x4_f32), a wave owning 16 rows x 192 columns of a slice:
//   pass 1 (along y), transposed:  t[c][y] = sum_r in[r][c] w[|r - (y - 32)|]     A[i = c][k = r]: for a k step the 16 lanes of a quarter
//            read 16 consecutive columns of ONE ray row; B[k = r][j = y] from the layer's weight table (64 floats of LDS per wave).
//            D[i = c][j = y]: lane = output row y, registers = 4 columns.
//   pass 2 (along x):  out[y][x] = sum_c t[c][y] w[|c - (x - 32)|]     A[i = y][k = c] IS pass 1's result where it lies: register s of
//            the lanes' own D is k step s. Pass 1 loads row i of its A from column 4 (i % 4) + i / 4 of the block (a permutation
//            within the same 64 bytes), so that D's register s of quarter kq is column 4 s + kq: a k step is four CONSECUTIVE columns,
//            and the steps outside an output block's reach are skipped. No transposition, no LDS round trip. B[k = c][j = x] from the
//            weight table again. D[i = y][j = x]: stored as 64-byte rows.
// The weights (the same pixel integrals as the general kernels: Taylor series for sigma >= 1.4 px, erf differences below) depend on the
// layer only: a wave looks its B operands up once per layer (12 + 20 registers), not per product. The layers of a slice are added in
// ascending order into the same accumulators (12 column blocks): reproducible.
//   k_superpose_uniform3  block = ONE slice, wave = one of its (<= 16) row blocks; the slice's layers staged in two LDS buffers a layer
//                         ahead, one barrier per layer. Ray grids of up to 128 columns (the reference's water cube: 0.24 ms).
//   k_superpose_uniform2  wave = (slice, row block, strip of 192 columns), A operands straight from global memory (the input rows are
//                         re-read by the row blocks within reach, through L1 / L2), no barrier in the loop: any grid (0.32 ms there — a
//                         wave waits a memory round trip per input column block).
// History on the reference's water cube (256^3, 20 layers; the general kernel of round 2 took 1.18 ms): vector ALUs with a sliding window
// 0.71 ms; first matrix version 0.67 ms; round 2's kernel — x pass, barrier, y pass, barrier per 32-row chunk staged in LDS, four waves
// per block, matrix cores busy 19 % — 0.36 ms; these two 0.32 and 0.24 ms. What is left is mostly matrix work: at the cube's radii
// (up to 16) a wave issues ~250 MFMAs per layer, two thirds of them in pass 2 (12 output column blocks against 8 input ones).
#pragma once
#include "rtd_kernels.hpp"

namespace rtd {

constexpr int kU2XB = 12;                            // output column blocks per wave (192 columns: the padded BEV of a 128-ray-wide grid)
constexpr int kU2Reach = 5;                          // input column blocks within reach of an output block: cb - 4 .. cb at radius 32 (ray column = BEV column - 32)
constexpr int kU2CB = kU2XB + kU2Reach - 1;          // input column blocks within reach of a strip
constexpr int kU2KS = (16 + 2 * kMaxSuperpR + 3) / 4 + 1;    // 21: k steps of pass 1 at the largest radius (rows within reach of a row block, from a multiple of four)
constexpr int kU2Run = 12;                           // ... of which the first 12 (radii up to 14) are requested together
constexpr int kU2Guard = 63;                         // weight table entry that is always zero

__global__ __launch_bounds__(256, 2) void k_superpose_uniform2(const float* __restrict__ bevIdd, const LayerPlan* __restrict__ layers,
                                                                const FieldState* __restrict__ st, FieldConst fc,
                                                                const unsigned int* __restrict__ sigMin, const float* __restrict__ stepTab,
                                                                float* __restrict__ bevDose, int nXS) {
    if (!st->uniformField || st->errorFlags) return;
    __shared__ float sW[4][64];                                      // per wave: w[0 .. rho] of its current layer, zeros beyond
    __shared__ int sLay[256], sRho[256], sCount[4];
    __shared__ float sRs[256];
    const int t = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, kq = lane >> 4;
    const int W = fc.W, H = fc.H, S = fc.S;
    const int first = st->beamFirstInside, passive = st->firstCalculatedPassive;
    const int nYB = (fc.bevH + 15) / 16, nParts = (nYB + 3) / 4;
    int item = blockIdx.x;
    const int part = item % nParts; item /= nParts;
    const int xs = item % nXS; item /= nXS;
    const int k = item;
    if (k < first || k >= passive) return;
    const size_t memStep = (size_t)W * H;
    // ---- the slice's depositing layers, ascending, with their 1/sigma and batch radius (thread l looks at layer l) ----
    {
        bool on = false; float rs = 0.0f; int rho = 0;
        if (t < fc.L && k < layers[t].layerFirstPassive) {
            const unsigned int bits = sigMin[(size_t)t * S + k];
            if (bits != 0x7f800000u) {                                // (+inf as stored by the reset: no live ray in this slice)
                const float sig2 = __uint_as_float(bits);
                const float sqrt2 = 1.41421356f;
                // 1/sigma of the slice's rays with k_fill's operations (same bits as its per-ray values); class and batch radius as there
                rs = stepTab[2 * k] * __builtin_amdgcn_rcpf(sqrt2 * (__builtin_amdgcn_sqrtf(sig2) + 0.21f));
                const float minRs = stepTab[2 * k] / (sqrt2 * (sqrtf(sig2) + 0.21f));
                int cls = f2iSat(fc.ksSigmaCutoff / (sqrtf(2.0f) * minRs) + 0.5f);
                cls = cls > kMaxSuperpR ? kMaxSuperpR : (cls < 0 ? 0 : cls);
                rho = layers[t].effRad[cls];
                on = true;
            }
        }
        const unsigned long long mask = __ballot(on);
        if (lane == 0) sCount[wv] = __popcll(mask);
        __syncthreads();
        int pos = __popcll(mask & ((1ull << lane) - 1ull));
        for (int w2 = 0; w2 < wv; ++w2) pos += sCount[w2];
        if (on) { sLay[pos] = t; sRho[pos] = rho; sRs[pos] = rs; }
        __syncthreads();
    }
    const int nA = sCount[0] + sCount[1] + sCount[2] + sCount[3];
    const int yb = 4 * part + wv;
    if (yb >= nYB) return;                                           // (no barrier below)
    const int y0 = 16 * yb;                                          // first padded-BEV row of the wave
    const int x0 = 16 * kU2XB * xs;                                  // first padded-BEV column of the strip
    const int cbBase = kU2XB * xs - (kU2Reach - 1);                               // input column block of index j = 0 (column block cb <-> ray columns 16 cb ...)
    const int nCB = W / 16;
    float* sw = sW[wv];

    f32x4 acc[kU2XB];
#pragma unroll
    for (int q = 0; q < kU2XB; ++q) acc[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    for (int a = 0; a < nA; ++a) {
        const int rho = sRho[a];
        const float rs = sRs[a];
        // rows of the slice within reach of the wave's 16 output rows (ray row = BEV row - 32), from a multiple of four
        const int rLo = max(0, y0 - 32 - rho) & ~3, rHi = min(H - 1, y0 - 17 + rho);
        if (rHi < rLo) continue;                                     // the layer deposits nothing in these rows (wave-uniform)
        const int nK = (rHi - rLo) / 4 + 1;                          // <= 21
        // ---- the layer's weights: lane i holds w[i] (pixel integrals as in k_superpose_mfma: Taylor series for sigma >= 1.4 px, erf below) ----
        {
            float w = 0.0f;
            if (lane <= rho) {
                const int ii = lane;
                if (rs <= 0.5f) {
                    const float h2 = rs * rs, h4 = h2 * h2;
                    const float k1 = h2 * (1.0f / 24.0f), k2 = h4 * (1.0f / 1920.0f), k3 = h4 * h2 * (1.0f / 322560.0f);
                    const float c0 = 1.0f - 2.0f * k1 + 12.0f * k2 - 120.0f * k3;
                    const float c1 = (4.0f * k1 - 48.0f * k2 + 720.0f * k3) * h2;
                    const float c2 = (16.0f * k2 - 480.0f * k3) * h4;
                    const float c3 = 64.0f * k3 * (h4 * h2);
                    const float wq = (float)(ii * ii);
                    const float gq = 0.5641895835f * rs * __builtin_amdgcn_exp2f(-1.4426950409f * h2 * wq);
                    w = gq * __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c3, wq, c2), wq, c1), wq, c0);
                } else {
                    w = ii == 0 ? erff(rs * 0.5f) : 0.5f * (erff(rs * ((float)ii + 0.5f)) - erff(rs * ((float)ii - 0.5f)));
                }
            }
            __builtin_amdgcn_wave_barrier();                         // (the previous layer's lookups are done: same wave, program order)
            sw[lane] = w;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const float* __restrict__ in = bevIdd + ((size_t)sLay[a] * S + k) * memStep;
        // The B operands depend on the layer only, not on the column block: looked up once per layer.
        //   pass 1: B[k = r][j = y] = w[|r - (y - 32)|], r = rLo + 4 s + kq, y = y0 + li                    -> b1[s]
        //   pass 2: B[k = c][j = x] = w[|c - (x - 32)|], c = 16 cb + 4 kq + s, x = 16 (cb + 4 - jj) + li      -> b2[jj][s]
        //           (input block cb against the output block jj blocks to its left of the farthest one it reaches)
        float b1[kU2KS], b2[kU2Reach][4];
        {
            const int d1 = rLo + kq - (y0 + li - 32);
#pragma unroll
            for (int s = 0; s < kU2KS; ++s) {
                const int d = d1 + 4 * s;
                b1[s] = s < nK ? sw[min(d < 0 ? -d : d, kU2Guard)] : 0.0f;
            }
            const int d2 = kq - li - 32;
#pragma unroll
            for (int jj = 0; jj < kU2Reach; ++jj)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const int d = d2 + 16 * jj + 4 * s2;
                    b2[jj][s2] = sw[min(d < 0 ? -d : d, kU2Guard)];
                }
        }
        // ---- per input column block within reach of the strip: pass 1, then at once pass 2 from its result into the output blocks
        //      within reach of it (q = j - 4 .. j) — one t block lives at a time ----
#pragma unroll
        for (int j = 0; j < kU2CB; ++j) {
            const int cb = cbBase + j;
            // needed by some output block of the strip: ray columns [x0 - 32 - rho, x0 + 16 kU2XB - 17 + rho]
            const bool need = cb >= 0 && cb < nCB && 16 * cb + 15 >= x0 - 32 - rho && 16 * cb <= x0 + 16 * kU2XB - 17 + rho;
            if (!need) continue;                                     // (wave-uniform)
            f32x4 tmp = {0.0f, 0.0f, 0.0f, 0.0f};
            const float* col = in + 16 * cb + 4 * (li & 3) + (li >> 2);   // row i of A <-> column 4 (i % 4) + i / 4 of the block (see pass 2)
            // pass 1: the first kU2Run k steps (radii up to 14 need no more) requested together, the rest when the radius asks
            {
                float av[kU2Run];
#pragma unroll
                for (int s1 = 0; s1 < kU2Run; ++s1) {
                    const int r = rLo + 4 * s1 + kq;
                    av[s1] = (s1 < nK && r < H) ? col[(size_t)r * W] : 0.0f;
                }
#pragma unroll
                for (int s1 = 0; s1 < kU2Run; ++s1)
                    if (s1 < nK) tmp = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s1], b1[s1], tmp, 0, 0, 0);
            }
            if (nK > kU2Run) {
                float av[kU2KS - kU2Run];
#pragma unroll
                for (int s1 = kU2Run; s1 < kU2KS; ++s1) {
                    const int r = rLo + 4 * s1 + kq;
                    av[s1 - kU2Run] = (s1 < nK && r < H) ? col[(size_t)r * W] : 0.0f;
                }
#pragma unroll
                for (int s1 = kU2Run; s1 < kU2KS; ++s1)
                    if (s1 < nK) tmp = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s1 - kU2Run], b1[s1], tmp, 0, 0, 0);
            }
            // pass 2: out[y][x] += sum_c t[c][y] w[|c - (x - 32)|]
#pragma unroll
            for (int jj = 0; jj < kU2Reach; ++jj) {
                const int q = j - jj;                                // (static)
                if (q < 0 || q >= kU2XB) continue;
                const int xq = x0 + 16 * q;                          // first padded-BEV column of the output block
                // ray columns of the input block [16 cb, 16 cb + 15] against the block's reach [xq - 32 - rho, xq - 17 + rho]
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)                       // k step s2 = the block's columns 4 s2 .. 4 s2 + 3: only those within the reach
                    if (16 * cb + 4 * s2 + 3 >= xq - 32 - rho && 16 * cb + 4 * s2 <= xq - 17 + rho)      // (wave-uniform)
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(tmp[s2], b2[jj][s2], acc[q], 0, 0, 0);
            }
        }
    }
    // ---- D[i = y][j = x]: lane = column, registers = rows 4 kq + reg ----
    float* out = bevDose + (size_t)k * fc.bevW * fc.bevH;
#pragma unroll
    for (int q = 0; q < kU2XB; ++q) {
        const int ox = x0 + 16 * q + li;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int oy = y0 + 4 * kq + reg;
            if (oy < fc.bevH && ox < fc.bevW) out[(size_t)oy * fc.bevW + ox] = acc[q][reg];
        }
    }
}

// The same two passes with the slice's layers staged in LDS: the block stages layer a + 1 (float4 loads, a layer ahead in registers,
// then one of two LDS buffers) while layer a is on the matrix cores: ONE barrier per layer, pass 1's A operands come from LDS (pitch
// W + 16: the four rows of a k step fall into different banks).
constexpr int kU3MaxV4 = 6;                          // float4 a thread stages per layer (H W / 4 <= 6 x 64 x row blocks)
template <int kMaxThreads>                           // 768: slices of up to 12 row blocks (three waves per SIMD: 168 registers); 1024: up to 16
__global__ __launch_bounds__(kMaxThreads) void k_superpose_uniform3(const float* __restrict__ bevIdd, const LayerPlan* __restrict__ layers,
                                                              const FieldState* __restrict__ st, FieldConst fc,
                                                              const unsigned int* __restrict__ sigMin, const float* __restrict__ stepTab,
                                                              float* __restrict__ bevDose) {
    if (!st->uniformField || st->errorFlags) return;
    extern __shared__ float sIn[];                                   // two buffers [H][W + 16]
    __shared__ float sW[16][64];                                     // per wave: w[0 .. rho] of its current layer, zeros beyond
    __shared__ int sLay[256], sRho[256], sCount[4];
    __shared__ float sRs[256];
    const int t = threadIdx.x, nT = blockDim.x;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, kq = lane >> 4;
    const int W = fc.W, H = fc.H, S = fc.S;
    const int first = st->beamFirstInside, passive = st->firstCalculatedPassive;
    const int k = blockIdx.x;
    if (k < first || k >= passive) return;
    const size_t memStep = (size_t)W * H;
    const int pitch = W + 16;
    // ---- the slice's depositing layers, ascending, with their 1/sigma and batch radius (thread l looks at layer l; L <= 256 <= nT) ----
    {
        bool on = false; float rs = 0.0f; int rho = 0;
        if (t < fc.L && k < layers[t].layerFirstPassive) {
            const unsigned int bits = sigMin[(size_t)t * S + k];
            if (bits != 0x7f800000u) {
                const float sig2 = __uint_as_float(bits);
                const float sqrt2 = 1.41421356f;
                rs = stepTab[2 * k] * __builtin_amdgcn_rcpf(sqrt2 * (__builtin_amdgcn_sqrtf(sig2) + 0.21f));
                const float minRs = stepTab[2 * k] / (sqrt2 * (sqrtf(sig2) + 0.21f));
                int cls = f2iSat(fc.ksSigmaCutoff / (sqrtf(2.0f) * minRs) + 0.5f);
                cls = cls > kMaxSuperpR ? kMaxSuperpR : (cls < 0 ? 0 : cls);
                rho = layers[t].effRad[cls];
                on = true;
            }
        }
        const unsigned long long mask = __ballot(on);
        if (lane == 0 && wv < 4) sCount[wv] = __popcll(mask);
        __syncthreads();
        int pos = __popcll(mask & ((1ull << lane) - 1ull));
        for (int w2 = 0; w2 < wv && w2 < 4; ++w2) pos += sCount[w2];
        if (on) { sLay[pos] = t; sRho[pos] = rho; sRs[pos] = rs; }
        __syncthreads();
    }
    const int nA = sCount[0] + sCount[1] + sCount[2] + sCount[3];
    const int y0 = 16 * wv;                                          // first padded-BEV row of the wave
    const int nCB = W / 16;
    float* sw = sW[wv];
    const int nV4 = (int)(memStep / 4), rowV4 = W / 4;

    // staging: this thread's float4 of a layer's slice -> registers (s0 .. s5) -> LDS buffer
    float4 s0, s1, s2, s3, s4, s5;
    s0 = s1 = s2 = s3 = s4 = s5 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    static_assert(kU3MaxV4 == 6, "six staging registers");
#define RTD_U3_FETCH(a_)                                                                                                     \
    {                                                                                                                        \
        const float4* __restrict__ src = reinterpret_cast<const float4*>(bevIdd + ((size_t)sLay[a_] * S + k) * memStep);     \
        if (t < nV4) s0 = src[t];                                                                                            \
        if (t + nT < nV4) s1 = src[t + nT];                                                                                  \
        if (t + 2 * nT < nV4) s2 = src[t + 2 * nT];                                                                          \
        if (t + 3 * nT < nV4) s3 = src[t + 3 * nT];                                                                          \
        if (t + 4 * nT < nV4) s4 = src[t + 4 * nT];                                                                          \
        if (t + 5 * nT < nV4) s5 = src[t + 5 * nT];                                                                          \
    }
    auto put = [&](float* dst, int i, const float4& v) {
        if (i < nV4) {
            const int r = i / rowV4, c = 4 * (i - r * rowV4);
            *reinterpret_cast<float4*>(dst + (size_t)r * pitch + c) = v;
        }
    };
#define RTD_U3_STORE(buf_)                                                                                                   \
    {                                                                                                                        \
        float* dst = sIn + (size_t)(buf_) * H * pitch;                                                                       \
        put(dst, t, s0); put(dst, t + nT, s1); put(dst, t + 2 * nT, s2); put(dst, t + 3 * nT, s3); put(dst, t + 4 * nT, s4); put(dst, t + 5 * nT, s5); \
    }

    f32x4 acc[kU2XB];
#pragma unroll
    for (int q = 0; q < kU2XB; ++q) acc[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    if (nA > 0) { RTD_U3_FETCH(0); RTD_U3_STORE(0); }
    __syncthreads();
    if (nA > 1) RTD_U3_FETCH(1);
    for (int a = 0; a < nA; ++a) {
        const int rho = sRho[a];
        const float rs = sRs[a];
        const float* __restrict__ in = sIn + (size_t)(a & 1) * H * pitch;
        const int rLo = max(0, y0 - 32 - rho) & ~3, rHi = min(H - 1, y0 - 17 + rho);
        if (rHi >= rLo) {                                            // (wave-uniform; a wave out of the layer's reach only keeps the barriers)
            const int nK = (rHi - rLo) / 4 + 1;                      // <= 21
            {
                float w = 0.0f;
                if (lane <= rho) {
                    const int ii = lane;
                    if (rs <= 0.5f) {
                        const float h2 = rs * rs, h4 = h2 * h2;
                        const float k1 = h2 * (1.0f / 24.0f), k2 = h4 * (1.0f / 1920.0f), k3 = h4 * h2 * (1.0f / 322560.0f);
                        const float c0 = 1.0f - 2.0f * k1 + 12.0f * k2 - 120.0f * k3;
                        const float c1 = (4.0f * k1 - 48.0f * k2 + 720.0f * k3) * h2;
                        const float c2 = (16.0f * k2 - 480.0f * k3) * h4;
                        const float c3 = 64.0f * k3 * (h4 * h2);
                        const float wq = (float)(ii * ii);
                        const float gq = 0.5641895835f * rs * __builtin_amdgcn_exp2f(-1.4426950409f * h2 * wq);
                        w = gq * __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c3, wq, c2), wq, c1), wq, c0);
                    } else {
                        w = ii == 0 ? erff(rs * 0.5f) : 0.5f * (erff(rs * ((float)ii + 0.5f)) - erff(rs * ((float)ii - 0.5f)));
                    }
                }
                __builtin_amdgcn_wave_barrier();
                sw[lane] = w;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            float b1[kU2Run], b2[kU2Reach][4];                       // (the k steps beyond kU2Run — radii above 14 — look their weights up on the spot)
            const int d1 = rLo + kq - (y0 + li - 32);
            {
#pragma unroll
                for (int s = 0; s < kU2Run; ++s) {
                    const int d = d1 + 4 * s;
                    b1[s] = s < nK ? sw[min(d < 0 ? -d : d, kU2Guard)] : 0.0f;
                }
                const int d2 = kq - li - 32;
#pragma unroll
                for (int jj = 0; jj < kU2Reach; ++jj)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        const int d = d2 + 16 * jj + 4 * s2;
                        b2[jj][s2] = sw[min(d < 0 ? -d : d, kU2Guard)];
                    }
            }
#pragma unroll
            for (int j = 0; j < kU2CB; ++j) {
                const int cb = j - (kU2Reach - 1);                   // (one strip: x0 = 0)
                const bool need = cb >= 0 && cb < nCB && 16 * cb + 15 >= -32 - rho && 16 * cb <= 16 * kU2XB - 17 + rho;
                if (!need) continue;                                 // (wave-uniform)
                f32x4 tmp = {0.0f, 0.0f, 0.0f, 0.0f};
                const float* col = in + 16 * cb + 4 * (li & 3) + (li >> 2);   // row i of A <-> column 4 (i % 4) + i / 4 of the block (see pass 2)
                {
                    float av[kU2Run];
#pragma unroll
                    for (int s1 = 0; s1 < kU2Run; ++s1) {
                        const int r = rLo + 4 * s1 + kq;
                        av[s1] = (s1 < nK && r < H) ? col[r * pitch] : 0.0f;
                    }
#pragma unroll
                    for (int s1 = 0; s1 < kU2Run; ++s1)
                        if (s1 < nK) tmp = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s1], b1[s1], tmp, 0, 0, 0);
                }
                if (nK > kU2Run) {
                    float av[kU2KS - kU2Run];
#pragma unroll
                    for (int s1 = kU2Run; s1 < kU2KS; ++s1) {
                        const int r = rLo + 4 * s1 + kq;
                        av[s1 - kU2Run] = (s1 < nK && r < H) ? col[r * pitch] : 0.0f;
                    }
#pragma unroll
                    for (int s1 = kU2Run; s1 < kU2KS; ++s1)
                        if (s1 < nK) {
                            const int d = d1 + 4 * s1;
                            tmp = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s1 - kU2Run], sw[min(d < 0 ? -d : d, kU2Guard)], tmp, 0, 0, 0);
                        }
                }
#pragma unroll
                for (int jj = 0; jj < kU2Reach; ++jj) {
                    const int q = j - jj;                            // (static)
                    if (q < 0 || q >= kU2XB) continue;
                    const int xq = 16 * q;
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)                   // k step s2 = the block's columns 4 s2 .. 4 s2 + 3: only those within the reach
                        if (16 * cb + 4 * s2 + 3 >= xq - 32 - rho && 16 * cb + 4 * s2 <= xq - 17 + rho)      // (wave-uniform)
                            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(tmp[s2], b2[jj][s2], acc[q], 0, 0, 0);
                }
            }
        }
        // layer a + 1 (in registers since the previous iteration) -> the other buffer, which every wave has left (barrier of the previous
        // iteration); then layer a + 2 is requested
        if (a + 1 < nA) RTD_U3_STORE((a + 1) & 1);
        __syncthreads();
        if (a + 2 < nA) RTD_U3_FETCH(a + 2);
    }
    float* out = bevDose + (size_t)k * fc.bevW * fc.bevH;
#pragma unroll
    for (int q = 0; q < kU2XB; ++q) {
        const int ox = 16 * q + li;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int oy = y0 + 4 * kq + reg;
            if (oy < fc.bevH && ox < fc.bevW) out[(size_t)oy * fc.bevW + ox] = acc[q][reg];
        }
    }
}

#undef RTD_U3_FETCH
#undef RTD_U3_STORE

}  // namespace rtd
