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
// layer only: a wave looks its B operands up once per layer, not per product. The layers of a slice are added in
// ascending order into the same accumulators (12 column blocks): reproducible.
//   k_superpose_uniform4  block = four row blocks of a slice, one per wave; the layers' rows within the block's reach staged in LDS a layer
//                         ahead. Ray grids of up to 128 columns (the reference's water cube: 0.17 ms).
//   k_superpose_uniform2  wave = (slice, row block, strip of 192 columns), A operands straight from global memory (the input rows are
//                         re-read by the row blocks within reach, through L1 / L2), no barrier in the loop: any grid (0.32 ms there — a
//                         wave waits a memory round trip per input column block).
// History on the reference's water cube (256^3, 20 layers; the general kernel of round 2 took 1.18 ms): vector ALUs with a sliding window
// 0.71 ms; first matrix version 0.67 ms; round 2's kernel — x pass, barrier, y pass, barrier per 32-row chunk staged in LDS, four waves
// per block, matrix cores busy 19 % — 0.36 ms; k_superpose_uniform2 0.32 ms; a block per slice with its layers double-buffered in LDS
// (k_superpose_uniform3, round 3, since removed) 0.24 ms: 190 of 256 CUs, and the counters showed three scalar branches per matrix
// instruction (the reach tests) — 0.225 ms with the radius classes below; the slice cut into three blocks 0.19 ms; reads first and
// selects after instead of predicated reads (a branch each) 0.18 ms; the next column block's values read ahead 0.17 ms. Clock stamps
// (tools/uniform_dbg.py): of ~15 000 cycles per layer and block, ~4000 are the two barriers and the store of the next layer's rows,
// ~5000 the issue of the thirteen loads of the layer after (the CU's eight waves queue 104 KB of requests at once), the matrix
// instructions of a wave ~4000 (two waves per SIMD). Spreading the loads over the column blocks moved the wait into the store
// (0.19 ms, more registers): the layer's rows arrive at ~2.4 TB/s over all CUs, twice the unique bytes (the blocks' reaches overlap).
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

// The same two passes with the layers' rows staged in LDS (pass 1's A operands come from LDS, pitch W + 16: the four rows of a k step
// fall into different banks), for ray grids of up to 128 columns. A block = kU4RB row blocks of one slice (the reference's water cube:
// three blocks of four per slice, 570 live blocks, two per CU at two waves per SIMD — a block per slice keeps 190 of the 256 CUs busy
// and the slices crossed by every layer set the time; the blocks beyond the 512 resident ones are the deep slices with few layers).
// A block stages, layer by layer, only the ray rows within reach of its own row blocks (<= kU4Rows for radii up to 16: 58 KB; a layer
// with a larger radius is read straight from global memory), a layer ahead through registers into one LDS buffer.
constexpr int kU4TabLayers = 32;
// w[ii] of a layer with 1 / (sqrt 2 sigma) = rs in ray units: the integral of the Gaussian over pixel ii (series for narrow pixels, as k_superpose_uniform2)
__device__ __forceinline__ float u4Weight(int ii, int rho, float rs) {
    float w = 0.0f;
    if (ii <= rho) {
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
    return w;
}
constexpr int kU4Waves = 4, kU4RB = 4, kU4Rows = 100, kU4Slack = 4, kU4StageV4 = 13;
__global__ __launch_bounds__(64 * kU4Waves, 2) void k_superpose_uniform4(const float* __restrict__ bevIdd, const LayerPlan* __restrict__ layers,
                                                              const FieldState* __restrict__ st, FieldConst fc,
                                                              const unsigned int* __restrict__ sigMin, const float* __restrict__ stepTab,
                                                              float* __restrict__ bevDose, long long* __restrict__ dbg) {
    if (!st->uniformField || st->errorFlags) return;
    const long long dbgT0 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;   // (RTD_UNIFORM_DEBUG: clock stamps, tools/uniform_dbg.py)
    long long dbgT1 = 0, dbgT2 = 0, dbgT3 = 0, dbgWork = 0, dbgBar = 0;
    extern __shared__ float sIn[];                                   // [kU4Rows][W + 16]: the ray rows within reach of the block's row blocks, one layer at a time
    __shared__ float sWall[kU4TabLayers][64];                        // the weights of the slice's first kU4TabLayers layers
    __shared__ float sW[kU4Waves][64];                                     // per wave: w[0 .. rho] of its current layer, zeros beyond
    __shared__ int sLay[256], sRho[256], sCount[4];
    __shared__ float sRs[256];
    const int t = threadIdx.x, nT = blockDim.x;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, kq = lane >> 4;
    const int W = fc.W, H = fc.H, S = fc.S;
    const int first = st->beamFirstInside, passive = st->firstCalculatedPassive;
    const int nYB = (fc.bevH + 15) / 16, nParts = (nYB + kU4RB - 1) / kU4RB;
    const int k = blockIdx.x / nParts, part = blockIdx.x % nParts;
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
    if (dbg) dbgT1 = (long long)__builtin_amdgcn_s_memtime();
    // ---- the layers' weights w[0 .. rho] (zeros beyond), once per block and all at once: the twelve waves of a slice would each run the
    //      same erf chain at the head of every layer ----
    for (int idx = t; idx < min(nA, kU4TabLayers) * 64; idx += 64 * kU4Waves) sWall[idx >> 6][idx & 63] = u4Weight(idx & 63, sRho[idx >> 6], sRs[idx >> 6]);
    __syncthreads();
    const int yb = kU4RB * part + wv;                                // the wave's row block
    const bool mine = wv < kU4RB && yb < nYB;
    const int y0 = 16 * yb;                                          // first padded-BEV row of the wave
    const int yFirst = 16 * kU4RB * part, yLast = min(16 * (kU4RB * part + kU4RB) - 1, 16 * nYB - 1);   // the block's rows
    const int nCB = W / 16;
    float* sWown = sW[wv];
    const int rowV4 = W / 4;

    f32x4 acc[kU2XB];
#pragma unroll
    for (int q = 0; q < kU2XB; ++q) acc[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- a layer's rows within reach of the block are [rB0, rB1]; a layer whose radius is beyond the buffer (rho > 16: more than kU4Rows
    //      rows) is read straight from global memory instead. The rows travel a layer ahead: fetched into registers before the arithmetic
    //      of the layer in front, stored to the (single) LDS buffer after it ----
    float4 s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12;                                                   // (scalars, not an array: an array indexed from lambdas goes to scratch)
    s0 = s1 = s2 = s3 = s4 = s5 = s6 = s7 = s8 = s9 = s10 = s11 = s12 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    static_assert(kU4StageV4 == 13, "thirteen staging registers");
    const int stepR = (64 * kU4Waves) / rowV4, stepC = (64 * kU4Waves) % rowV4;
    const int sr0 = t / rowV4, sc0 = t - sr0 * rowV4;
    auto reach = [&](int a2, int& rB0, int& rB1) {
        const int rho2 = sRho[a2];
        rB0 = max(0, yFirst - 32 - rho2) & ~3; rB1 = min(H - 1, yLast - 32 + rho2);
    };
#define RTD_U4_FETCH(a_)                                                                                                     \
    {                                                                                                                        \
        int rB0_, rB1_; reach(a_, rB0_, rB1_);                                                                               \
        const int nRows_ = rB1_ - rB0_ + 1;                                                                                  \
        if (nRows_ <= kU4Rows && nRows_ > 0) {                                /* (block-uniform) */                          \
            const float4* __restrict__ src = reinterpret_cast<const float4*>(bevIdd + ((size_t)sLay[a_] * S + k) * memStep + (size_t)rB0_ * W); \
            const int nV_ = nRows_ * rowV4;                                   /* <= kU4Rows * 32 <= kU4StageV4 * 256 */      \
            if (t + 0 * 64 * kU4Waves < nV_) s0 = src[t + 0 * 64 * kU4Waves];                                                          \
            if (t + 1 * 64 * kU4Waves < nV_) s1 = src[t + 1 * 64 * kU4Waves];                                                          \
            if (t + 2 * 64 * kU4Waves < nV_) s2 = src[t + 2 * 64 * kU4Waves];                                                          \
            if (t + 3 * 64 * kU4Waves < nV_) s3 = src[t + 3 * 64 * kU4Waves];                                                          \
            if (t + 4 * 64 * kU4Waves < nV_) s4 = src[t + 4 * 64 * kU4Waves];                                                          \
            if (t + 5 * 64 * kU4Waves < nV_) s5 = src[t + 5 * 64 * kU4Waves];                                                          \
            if (t + 6 * 64 * kU4Waves < nV_) s6 = src[t + 6 * 64 * kU4Waves];                                                          \
            if (t + 7 * 64 * kU4Waves < nV_) s7 = src[t + 7 * 64 * kU4Waves];                                                          \
            if (t + 8 * 64 * kU4Waves < nV_) s8 = src[t + 8 * 64 * kU4Waves];                                                          \
            if (t + 9 * 64 * kU4Waves < nV_) s9 = src[t + 9 * 64 * kU4Waves];                                                          \
            if (t + 10 * 64 * kU4Waves < nV_) s10 = src[t + 10 * 64 * kU4Waves];                                                        \
            if (t + 11 * 64 * kU4Waves < nV_) s11 = src[t + 11 * 64 * kU4Waves];                                                        \
            if (t + 12 * 64 * kU4Waves < nV_) s12 = src[t + 12 * 64 * kU4Waves];                                                        \
        }                                                                                                                    \
    }
    int pr, pc;
    auto put = [&](int i, int nV2, const float4& v) {
        if (i < nV2) *reinterpret_cast<float4*>(sIn + pr * pitch + 4 * pc) = v;
        pr += stepR; pc += stepC;
        if (pc >= rowV4) { pc -= rowV4; ++pr; }
    };
    if (dbg) dbgT2 = (long long)__builtin_amdgcn_s_memtime();
    if (nA > 0) RTD_U4_FETCH(0);
    for (int a = 0; a < nA; ++a) {
        const int rho = sRho[a];
        const float rs = sRs[a];
        int rB0, rB1; reach(a, rB0, rB1);
        const int nRowsB = rB1 - rB0 + 1;
        const bool staged = nRowsB <= kU4Rows;                       // (block-uniform)
        const float* __restrict__ gIn = bevIdd + ((size_t)sLay[a] * S + k) * memStep;
        const long long db0 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
        __syncthreads();                                             // the previous layer's rows are consumed
        if (staged && nRowsB > 0) {
            const int nV = nRowsB * rowV4;
            pr = sr0; pc = sc0;
            put(t + 0 * 64 * kU4Waves, nV, s0);
            put(t + 1 * 64 * kU4Waves, nV, s1);
            put(t + 2 * 64 * kU4Waves, nV, s2);
            put(t + 3 * 64 * kU4Waves, nV, s3);
            put(t + 4 * 64 * kU4Waves, nV, s4);
            put(t + 5 * 64 * kU4Waves, nV, s5);
            put(t + 6 * 64 * kU4Waves, nV, s6);
            put(t + 7 * 64 * kU4Waves, nV, s7);
            put(t + 8 * 64 * kU4Waves, nV, s8);
            put(t + 9 * 64 * kU4Waves, nV, s9);
            put(t + 10 * 64 * kU4Waves, nV, s10);
            put(t + 11 * 64 * kU4Waves, nV, s11);
            put(t + 12 * 64 * kU4Waves, nV, s12);
        }
        __syncthreads();
        const long long db1 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
        if (a + 1 < nA) RTD_U4_FETCH(a + 1);
        const int rLo = max(0, y0 - 32 - rho) & ~3, rHi = min(H - 1, y0 - 17 + rho);
        // (two instances of the layer's arithmetic, so that the staged one reads with LDS instructions and not through a flat pointer)
        auto layer = [&](auto stagedTag) {
            constexpr bool kStaged = decltype(stagedTag)::value;
            const float* __restrict__ in = kStaged ? (const float*)sIn : gIn;   // row r of the slice at in[(r - rOff) * inPitch]
            const int inPitch = kStaged ? pitch : W;
            const int rOff = kStaged ? rB0 : 0;
            const int rMax = kStaged ? rB1 : H - 1;                  // (rows past it carry zero weights)
            const int nK = (rHi - rLo) / 4 + 1;                      // <= 21
            const float* sw = sWall[a];
            if (a >= kU4TabLayers) {                                 // (more layers through the slice than the table holds: on the spot)
                const float w = u4Weight(lane, rho, rs);
                __builtin_amdgcn_wave_barrier();
                sWown[lane] = w;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                sw = sWown;
            }
            const int d1 = rLo + kq - (y0 + li - 32);
            const int rHi4 = min(rMax, rLo + 4 * nK - 1);            // (the generic form's rows: s1 < nK and r <= rMax)
            // A radius class R (rho <= R <= 16) fixes the pattern of the layer's matrix instructions: (18 + 2 R) / 4 + 1 k steps per column
            // block in pass 1, the k steps 29 - R <= 16 jj + 4 s2 <= 47 + R in pass 2 — no branch per instruction (the counters showed
            // three scalar branches and nine scalar instructions per matrix instruction in the generic form below, the matrix pipe 20 % busy);
            // the k steps beyond the layer's own reach multiply zero weights.
            auto layerStatic = [&](auto rTag) {
                constexpr int R = decltype(rTag)::value, NK = (18 + 2 * R) / 4 + 1;
                float b1[NK], b2[kU2Reach][4];
#pragma unroll
                for (int s = 0; s < NK; ++s) {
                    const int d = d1 + 4 * s;
                    b1[s] = sw[min(d < 0 ? -d : d, kU2Guard)];
                }
                const int d2 = kq - li - 32;
#pragma unroll
                for (int jj = 0; jj < kU2Reach; ++jj)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        const int d = d2 + 16 * jj + 4 * s2;
                        if (16 * jj + 4 * s2 >= 29 - R && 16 * jj + 4 * s2 <= 47 + R) b2[jj][s2] = sw[min(d < 0 ? -d : d, kU2Guard)];
                    }
                const float* col0 = in + (rLo + kq - rOff) * inPitch + 4 * (li & 3) + (li >> 2);
                // (read first, choose then: a load under a condition becomes a branch; the rows past the staged ones lie within the kU4Slack
                //  rows of the buffer. The values of column block cb + 1 — of the pitch's padding after the last — are read ahead of cb's arithmetic)
                float av[NK], avN[NK];
#pragma unroll
                for (int s1 = 0; s1 < NK; ++s1) avN[s1] = col0[4 * s1 * inPitch];
#pragma unroll
                for (int j = kU2Reach - 1; j < kU2CB; ++j) {
                    const int cb = j - (kU2Reach - 1);
                    if (cb >= nCB) break;                            // (wave-uniform)
                    f32x4 tmp = {0.0f, 0.0f, 0.0f, 0.0f};
                    const float* colN = col0 + 16 * (cb + 1);
#pragma unroll
                    for (int s1 = 0; s1 < NK; ++s1) av[s1] = (rLo + 4 * s1 + kq <= rHi4) ? avN[s1] : 0.0f;
#pragma unroll
                    for (int s1 = 0; s1 < NK; ++s1) avN[s1] = colN[4 * s1 * inPitch];
#pragma unroll
                    for (int s1 = 0; s1 < NK; ++s1) tmp = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s1], b1[s1], tmp, 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < kU2Reach; ++jj) {
                        const int q = j - jj;                        // (static)
                        if (q < 0 || q >= kU2XB) continue;
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2)
                            if (16 * jj + 4 * s2 >= 29 - R && 16 * jj + 4 * s2 <= 47 + R)      // (static)
                                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(tmp[s2], b2[jj][s2], acc[q], 0, 0, 0);
                    }
                }
            };
            if (kStaged && rho <= 4) layerStatic(std::integral_constant<int, 4>{});
            else if (kStaged && rho <= 8) layerStatic(std::integral_constant<int, 8>{});
            else if (kStaged && rho <= 12) layerStatic(std::integral_constant<int, 12>{});
            else if (kStaged && rho <= 16) layerStatic(std::integral_constant<int, 16>{});
            else {
            float b1[kU2Run], b2[kU2Reach][4];                       // (the k steps beyond kU2Run look their weights up on the spot)
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
                        av[s1] = (s1 < nK && r <= rMax) ? col[(r - rOff) * inPitch] : 0.0f;
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
                        av[s1 - kU2Run] = (s1 < nK && r <= rMax) ? col[(r - rOff) * inPitch] : 0.0f;
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
        };
        if (mine && rHi >= rLo) {                                    // (wave-uniform; a wave out of the layer's reach only keeps the barriers)
            if (staged) layer(std::true_type{}); else layer(std::false_type{});
        }
        if (dbg) { dbgBar += db1 - db0; dbgWork += (long long)__builtin_amdgcn_s_memtime() - db1; }
    }
    if (dbg) dbgT3 = (long long)__builtin_amdgcn_s_memtime();
    float* out = bevDose + (size_t)k * fc.bevW * fc.bevH;
#pragma unroll
    for (int q = 0; q < kU2XB; ++q) {
        const int ox = 16 * q + li;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int oy = y0 + 4 * kq + reg;
            if (mine && oy < fc.bevH && ox < fc.bevW) out[(size_t)oy * fc.bevW + ox] = acc[q][reg];
        }
    }
    if (dbg && lane == 0) {
        long long* q = dbg + (size_t)blockIdx.x * 16;
        if (wv == 0) {
            q[0] = dbgT0; q[1] = dbgT1; q[2] = dbgT2; q[3] = dbgT3; q[4] = (long long)__builtin_amdgcn_s_memtime();
            q[5] = ((long long)k << 32) | (unsigned)part; q[6] = nA;
            q[7] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // XCC_ID | HW_ID
        }
        q[8 + wv] = dbgWork; q[12 + wv] = dbgBar;
    }
}

#undef RTD_U4_FETCH

}  // namespace rtd
