// rtd_sweep.hpp — K7s: the kernel superposition as a source-stationary row sweep on the matrix cores (gfx950, wave64).
//
// Same arithmetic as kernelSuperposition<rad> (reference src/kernel_wrapper.cuh:432-489): every source voxel s of a BEV slice adds
// the separable patch  dose_s * e_s[|dy|] * e_s[|dx|],  |dy|, |dx| <= rho_s  (rho_s = batch radius of its 32 x 8 tile, e_s = the
// erf-difference weights of ITS OWN 1/sigma, kernel_wrapper.cuh:459-467). k_superpose_mfma (rtd_kernels.hpp) owns OUTPUT tiles and
// visits, per 16 x 16 tile, every source quad whose patch touches it: 4 MFMAs per quad at rho ~ 8, ~13 vector + ~10 scalar
// instructions of address arithmetic per MFMA, and every source's weight table built by each of the ~1.9 output tiles it reaches
// (profiles/r02_pmc_sq_superpose.txt). This kernel turns the loops around:
//
//   * A patch is EVEN in dy. For ONE source row ys the contributions of all its sources — of all energy layers — to the output rows
//     ys + dy and ys - dy are the same numbers  T[|dy|][x] = sum_s (w_s e_s[|dy|]) (w_s e_s[|x - x_s|]),  w_s = sqrt(dose_s):
//     rho + 1 <= 16 distinct rows instead of 2 rho + 1. T is ONE 16-row MFMA tile per 16 output columns (v_mfma_f32_16x16x4_f32:
//     M = |dy|, N = x, K = four neighbouring sources of the row), accumulated over the layers in one chain and handed to the output
//     rows ys +- |dy| once per source ROW, not per (row, layer).
//   * The band structure is static. Source quad q and output column block t are a pair iff |16 (t - 1) - 4 q + (li - kq)| <= rho
//     for some lane: a compile-time set per radius level (pairs that come into reach at rho = 0, 1, 5, 9, 13). Fully unrolled; a pair
//     is two ds_read_b32 with immediate offsets + one MFMA. No per-visit address arithmetic, no masks, no ballots: the operand
//     addresses (table entry min(|d|, guard) of the lane's source) are 13 registers computed once per block, and a level's operands
//     are requested in one burst while the previous level is on the matrix cores.
//   * Every source's table is built exactly once per launch (a wave owns the 64 consecutive sources of a row), with w_s folded in
//     (w e[|dy|] * w e[|dx|] = dose e e: no per-operand multiply).
//
// Work item = block of 8 waves = (step k, 64 x 64 patch of sources, layer group g); wave w sweeps the patch rows w, w + 8, ...
// over the group's layers. The 96 x 96 output tile of the patch lives in LDS; a wave adds its T into the tile rows ys +- |dy| when
// the flush ticket (an LDS word) shows its source row: rows are flushed in ascending order, so every element receives its addends in
// a fixed order and the BEV dose stays bitwise reproducible like k_superpose_mfma's. Plain read-modify-write under the ticket (LDS
// float atomics measured 2x slower), tile pitch 100 (conflict-free for the MFMA result layout). The groups' (and neighbouring
// patches') tiles are summed by the LAST block of the step to arrive (agent-scope hand-off as in k_superpose_mfma's tree;
// MI355X_MICROARCH.md "Correctness boundaries"), which writes the whole BEV slice, zeros included. Tiles with a batch radius above 16
// (reference limit: 32; a few deep steps of the upper layers in the fields that have them at all) are left out here: the second launch,
// k_superpose_sweep_big (rtd_sweep_big.hpp), superposes exactly those and adds them to the slices; it returns at once when the
// field's largest batch radius (FieldState::maxRadius) shows that there are none.
// Measured and dropped (C3, parity-green): 16 waves per block that leave their T in LDS and, after a barrier, gather the rows of the
// tile they own (no ticket; tile in registers): one block per CU, so nothing overlaps the gather phases — 0.51 ms against 0.35 ms.
#pragma once
#include "rtd_kernels.hpp"

namespace rtd {

constexpr int kSwWaves = 8;                          // waves per block
constexpr int kSwPatch = 64;                         // source columns per patch (one lane per source of a row)
constexpr int kSwPatchRows = 64;                     // source rows per patch
constexpr int kSwMaxR = 16;                          // largest batch radius this kernel takes
constexpr int kSwOut = kSwPatch + 2 * kSwMaxR;       // 96: edge of the output tile of a patch
constexpr int kSwOutRows = kSwPatchRows + 2 * kSwMaxR;   // rows of the output tile
constexpr int kSwNCB = kSwOut / 16;                  // 6 column blocks
constexpr int kSwGuard = kSwMaxR + 1;                // table entry 17 is always zero: the lookups clamp to it
constexpr int kSwTS = 19;                            // floats per source table (entries 0 .. 17 + pad; odd: the build's stores are conflict-free)
constexpr int kSwPitch = 100;                        // row pitch of the tile in LDS: 4 rows = 400 floats = 16 banks on — the four row groups
                                                     // of a flush (rows 4 kq + r, columns li) fall into 64 different banks
constexpr int kSwSlot = kSwOutRows * kSwOut;             // floats of a partial tile
constexpr int kSwMaxLay = 64;                        // layers per group (one lane each when the list is made)
constexpr int kSwMaxGroups = 16;
constexpr int kSwTileRows = kSwPatchRows / 8 + 1, kSwTileCols = 3;      // 32 x 8 classification tiles a 64 x 64 patch can touch
constexpr int kSwTk = 3;                             // column blocks per flush ticket (two tickets: 0.309 ms; three tickets of two blocks 0.315, one ticket 0.311)
constexpr int kSwNDelta = 12;                        // pair offsets delta = -28, -24, ..., 16

// smallest radius at which quad offset delta = (first output column of the block) - (first source of the quad) pairs them:
// the lanes' distances are delta + (li - kq), li - kq in [-3, 15]
__host__ __device__ constexpr int swNeed(int delta) { return delta > 3 ? delta - 3 : (delta < -15 ? -delta - 15 : 0); }

// dynamic LDS (floats): the output tile [96][kSwPitch], then the weight tables [wave][64][kSwTS]
constexpr int kSwLdsTab = kSwOutRows * kSwPitch;
constexpr int kSwWaveLds = 64 * kSwTS;                    // a wave's tables
constexpr int kSwLdsWords = kSwLdsTab + kSwWaves * kSwWaveLds;

// erf(t), t >= 0, for the tables of sources sharper than sigma = 1.4 pixels: the two branches of rtd_erf_det (include/rtd_detmath.h:
// t + t R(t^2) below 0.875, 1 - 2^P(t - 0.875) above) evaluated side by side and selected — no divergence, the hardware exp2 for
// the power of two (this value is a weight, not index work). |absolute error| <= 1.5e-7; ~27 vector instructions against the
// ~70 (both sides of its branches) of the library erff this kernel first used.
__device__ inline float swErf(float t) {
    const float s2 = t * t;
    float r = 8.694667811e-05f;
    r = __builtin_fmaf(r, s2, -8.215559851e-04f);
    r = __builtin_fmaf(r, s2, 5.207134257e-03f);
    r = __builtin_fmaf(r, s2, -2.686173980e-02f);
    r = __builtin_fmaf(r, s2, 1.128373582e-01f);
    r = __builtin_fmaf(r, s2, -3.761263625e-01f);
    r = __builtin_fmaf(r, s2, 1.283791669e-01f);
    const float small = __builtin_fmaf(r, t, t);
    const float u = fminf(t, 4.0f) - 0.875f;                         // (erfc(4) < 2^-25: 1 beyond)
    float p = -3.116289875e-08f;
    p = __builtin_fmaf(p, u, 2.754377229e-06f);
    p = __builtin_fmaf(p, u, -5.187475768e-05f);
    p = __builtin_fmaf(p, u, 5.094422115e-04f);
    p = __builtin_fmaf(p, u, -3.331390714e-03f);
    p = __builtin_fmaf(p, u, 1.650326662e-02f);
    p = __builtin_fmaf(p, u, -6.772692889e-02f);
    p = __builtin_fmaf(p, u, -1.192432172e+00f);
    p = __builtin_fmaf(p, u, -3.506067286e+00f);
    p = __builtin_fmaf(p, u, -2.211398194e+00f);
    const float big = 1.0f - __builtin_amdgcn_exp2f(p);
    return t < 0.875f ? small : big;
}

// One (source row, layer): tables of the wave's 64 sources -> LDS. Entry i of a source = w * e_i for i <= its own batch radius, zero
// up to the largest index the previous row-layer of this wave wrote (the lookups never clamp per source, only to the guard).
// e_i: pixel-integrated Gaussian, the Taylor series of k_superpose_mfma for 1/sigma <= 0.5 (< 3e-8 absolute), erf differences above.
template <bool ALL>
__device__ inline void swBuildSeries(float* __restrict__ m, float rs, float w, int rhoS, int rhoRow, bool dead, bool series) {
    const float r = dead ? 0.25f : rs;                               // (a dead ray's 1/sigma is +inf: keep the arithmetic finite, the values are not stored)
    const float h2 = r * r, h4 = h2 * h2;
    const float k1 = h2 * (1.0f / 24.0f), k2 = h4 * (1.0f / 1920.0f), k3 = h4 * h2 * (1.0f / 322560.0f);
    const float c0 = 1.0f - 2.0f * k1 + 12.0f * k2 - 120.0f * k3;
    const float c1 = (4.0f * k1 - 48.0f * k2 + 720.0f * k3) * h2;
    const float c2 = (16.0f * k2 - 480.0f * k3) * h4;
    const float c3 = 64.0f * k3 * (h4 * h2);
    float q = __builtin_amdgcn_exp2f(-1.4426950409f * h2), gq = 0.5641895835f * r * w;   // gq = w * rs/sqrt(pi) * exp(-x_i^2)
    const float cq = q * q;
    if (ALL || series) m[0] = dead ? 0.0f : c0 * gq;
    gq *= q; q *= cq;
    for (int i = 1; i <= rhoRow; i += 2) {
        const float w0 = (float)(i * i), w1 = (float)((i + 1) * (i + 1));
        const float s0 = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c3, w0, c2), w0, c1), w0, c0);
        const float s1 = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c3, w1, c2), w1, c1), w1, c0);
        const float g1 = gq * q, q1 = q * cq;
        const float ea = gq * s0, eb = g1 * s1;
        gq = g1 * q1; q = q1 * cq;
        if (ALL || series) {
            m[i] = i <= rhoS ? ea : 0.0f;
            if (i + 1 <= rhoRow) m[i + 1] = i + 1 <= rhoS ? eb : 0.0f;
        }
    }
}

// One (source row, layer): tables of the wave's 64 sources -> LDS. Entry i of a source = w * e_i for i <= its own batch radius, zero
// up to the largest index the previous row-layer of this wave wrote (the lookups never clamp per source, only to the guard).
// e_i: pixel-integrated Gaussian, the Taylor series of k_superpose_mfma for 1/sigma <= 0.5 (< 3e-8 absolute), erf differences above.
__device__ inline void swBuild(float* __restrict__ m, float rs, float w, int rhoS, int rhoRow, int prevRho) {
    const bool dead = rhoS < 0;
    // Series up to 1/sigma = 0.75 (sigma >= 0.94 pixels): the first neglected term, H8(x) rs^8 / 92897280 relative to e_0, is
    // 1.8e-5 rs^8 = 1.8e-6 there (k_superpose_mfma stops at 0.5, 7e-8; the parity bar is 1e-4) — the erf form is left to the
    // sources of radius <= 3, a fifth of them instead of a third.
    const bool series = dead || rs <= 0.75f;
    if (__all(series)) swBuildSeries<true>(m, rs, w, rhoS, rhoRow, dead, true);      // (the usual case: no per-store predicate)
    else {
        if (__any(series)) swBuildSeries<false>(m, rs, w, rhoS, rhoRow, dead, series);
        // sources sharper than sigma = 1.4 pixels (few entries)
        const float r = series ? 1.0f : rs;                          // (finite arguments on the lanes whose values are not stored)
        float erfNew = swErf(r * 0.5f), erfOld = -erfNew;
        for (int i = 0; i <= rhoRow; ++i) {
            const float e = 0.5f * (erfNew - erfOld) * w;
            erfOld = erfNew;
            erfNew = swErf(r * ((float)i + 1.5f));
            if (!series) m[i] = i <= rhoS ? e : 0.0f;
        }
    }
    for (int i = rhoRow + 1; i <= prevRho; ++i) m[i] = 0.0f;         // what the previous row-layer left beyond this one's reach
}

// The pairs that come into reach at radius NEED (0, 1, 5, 9, 13): T += A(quad) * B(quad, column block). A row-layer of radius rho runs
// the levels NEED <= rho one after the other, each in runs of up to eight quads: a run's A and B operands are requested from LDS in
// one burst, then its MFMAs follow (sched_barrier between them) — an MFMA never waits for a read issued just in front of it; with
// 2..4 waves per SIMD that round trip per MFMA was what the first version of this kernel spent its time on. (Requesting the next
// level's operands while the previous level multiplies measured the same, 0.311 ms, at 124 instead of 99 registers.)
template <int NEED>
struct SwLevel {
    static constexpr int count() { int n = 0; for (int q = 0; q < 16; ++q) for (int t = 0; t < kSwNCB; ++t) n += swNeed(16 * (t - 1) - 4 * q) == NEED ? 1 : 0; return n; }
    static constexpr int kN = count();
};
// Operand addresses of the pair offsets: the four of level 0 (delta = -12 .. 0) live in registers for the whole block; a higher level
// has two offsets (delta = -15 - NEED and 3 + NEED: -16 / 4, -20 / 8, -24 / 12, -28 / 16), computed when the level runs (3 vector
// instructions each).
struct SwIdx {
    int base, lk;                                                    // the lane's source of quad 0; li - kq
    int d0[4];                                                       // delta = -12, -8, -4, 0
    __device__ inline int at(int delta) const { return base + min(abs(delta + lk), kSwGuard); }
};
// quads [Q0, Q1) of a level (a patch whose last columns lie outside the field's dose rectangle skips its last quads: their tables are zero)
template <int NEED, int Q0, int Q1>
__device__ inline void swLevelLoadQ(float (&b)[SwLevel<NEED>::kN], const float* __restrict__ lds, const SwIdx& ix) {
    const int lo = NEED == 0 ? 0 : ix.at(-15 - NEED), hi = NEED == 0 ? 0 : ix.at(3 + NEED);   // the level's two offsets
    int n = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int t = 0; t < kSwNCB; ++t) {
            const int delta = 16 * (t - 1) - 4 * q;                  // a constant once unrolled
            if (swNeed(delta) == NEED) {
                const int idx = NEED == 0 ? ix.d0[(delta + 12) >> 2] : (delta < 0 ? lo : hi);
                if (q >= Q0 && q < Q1) b[n] = lds[idx + q * 4 * kSwTS];
                ++n;
            }
        }
}
template <int NEED, int Q0, int Q1>
__device__ inline void swLevelMulQ(f32x4 (&acc)[kSwNCB], const float (&a)[16], const float (&b)[SwLevel<NEED>::kN]) {
    int n = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int t = 0; t < kSwNCB; ++t)
            if (swNeed(16 * (t - 1) - 4 * q) == NEED) {
                if (q >= Q0 && q < Q1) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[n], acc[t], 0, 0, 0);
                ++n;
            }
}
// One level for the quads [Q0, Q1): its A and B operands requested in one burst, then its MFMAs (registers: at most 2 x 8).
template <int NEED, int Q0, int Q1>
__device__ inline void swLevelRunQ(f32x4 (&acc)[kSwNCB], const float* __restrict__ lds, int idxA, const SwIdx& ix) {
    float a[16], b[SwLevel<NEED>::kN];
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        bool any = false;
#pragma unroll
        for (int t = 0; t < kSwNCB; ++t) any = any || swNeed(16 * (t - 1) - 4 * q) == NEED;
        if (any) a[q] = lds[idxA + q * 4 * kSwTS];
    }
    swLevelLoadQ<NEED, Q0, Q1>(b, lds, ix);
    __builtin_amdgcn_sched_barrier(0);
    swLevelMulQ<NEED, Q0, Q1>(acc, a, b);
}
constexpr int kSwQCut = 14;                                          // quads [0, 14) always, [14, 16) only when the patch has more than 56 columns
template <int NEED>
__device__ inline void swLevelRun(f32x4 (&acc)[kSwNCB], const float* __restrict__ lds, int idxA, const SwIdx& ix, bool tail) {
    swLevelRunQ<NEED, 0, 8>(acc, lds, idxA, ix);
    swLevelRunQ<NEED, 8, kSwQCut>(acc, lds, idxA, ix);
    if (tail) swLevelRunQ<NEED, kSwQCut, 16>(acc, lds, idxA, ix);
}
// SELF: the launch plans for itself — no k_ks_plan in front. Block 0 IS the plan (ksPlanBody: the state record, the batch radii, the
// boxes of the transfer — what the launches behind this one read), and every other block derives what IT needs from k_fill's results:
// the steps its group deposits at and the field's last one from the layers' layerFirstPassive, the batch radii of its own layers from
// their class histograms (batchRadii, the plan's own function), a radius overflow from the histograms' last bin. One launch and its
// gap less on the field's critical path (k_ks_plan 11.8 us + 4.7 us). The host chooses SELF once a finished compute has told it that
// the field is not a uniform-sigma one (the separable kernel sits between the plan and this launch otherwise); no NUCLEAR_CORR.
template <bool SELF>
__global__ __launch_bounds__(64 * kSwWaves, 4) void k_superpose_sweep(const float* __restrict__ bevIdd, const float* __restrict__ bevRSigmaEff,
                                                                   const unsigned char* __restrict__ tileRad, const LayerPlan* __restrict__ layers,
                                                                   const FieldState* __restrict__ st, FieldConst fc, int G, int nPXg, int nPYg,
                                                                   const int* __restrict__ active, float* __restrict__ slots, int* __restrict__ counters,
                                                                   float* __restrict__ bevDose, long long* __restrict__ dbg, const KsPlanArgs* __restrict__ kaSelf) {
    extern __shared__ float sw[];
    if (SELF && blockIdx.x == 0) {                                   // (dispatched first; its arguments in device memory: by value they
        //  would sit in every block's registers — or scratch — for nothing)
        ksPlanBody(*kaSelf, fc, (int)threadIdx.x, 64 * kSwWaves, *reinterpret_cast<KsPlanLds*>(sw));
        return;
    }
    // diagnostic build only (RTD_SWEEP_DEBUG): clock stamps per block — no output value depends on them
    const long long dbgT0 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    long long dbgT1 = 0, dbgT2 = 0, dbgT3 = 0;
    __shared__ signed char sEff[kSwMaxLay * kSwTileRows * kSwTileCols];   // [layer slot][9][3] batch radius of the tile (-1: none)
    __shared__ int sRows[kSwMaxLay];                                 // [layer slot]: first | last << 8 patch row that carries dose
    __shared__ int sLay[kSwMaxLay];                                  // [layer slot]: layer
    __shared__ int sMisc[4];                                         // number of layer slots, -, "last block of the step"
    __shared__ int sTicket[kSwNCB / kSwTk];                          // flush tickets: the source row whose turn it is, per group of column blocks
    float* sOut = sw;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- decode (block-uniform); the deepest steps first (measured: ascending order 0.342 ms against 0.313 ms) ----
    // (group fastest, then step, patch slowest: the live blocks — steps [entry, passive) of the patches that hold dose — are then
    //  CONSECUTIVE block indices, which the dispatcher deals round-robin to the 8 XCDs. With the patch index in the middle the
    //  live blocks of the bench field, one patch of four, all had indices 16 j + 0..3 and landed on four of the eight XCDs.)
    int item = SELF ? (int)blockIdx.x - 1 : (int)blockIdx.x;
    const int g = item % G; item /= G;
    const int nPg = nPXg * nPYg;
    const int k = fc.S - 1 - item % fc.S; item /= fc.S;
    const int p = item;
    __shared__ int sPlan[2 + kSwMaxGroups];                          // SELF: the field's last depositing step + 1, radius overflow, the groups' last steps + 1
    __shared__ unsigned short sLfp[SELF ? kMaxLayers : 1];           // SELF: layerFirstPassive of every layer
    int* sEffRad = reinterpret_cast<int*>(sw);                       // SELF: [layer of the group][34] batch radius per class, in the tile's LDS before it is zeroed
    if (SELF) {
        // one round trip for everything the block plans from: every layer's last step (threads 0 .. L - 1) and the class histograms of
        // the group's layers (threads 256 ...), batched with the plan's own function
        if (tid < 2 + kSwMaxGroups) sPlan[tid] = 0;
        __syncthreads();
        for (int l = tid; l < fc.L; l += 64 * kSwWaves) {
            const int lfp = layers[l].layerFirstPassive;
            sLfp[l] = (unsigned short)lfp;
            atomicMax(&sPlan[0], lfp);
            atomicMax(&sPlan[2 + l % G], lfp);
            if (layers[l].hist[kMaxSuperpR + 1] > 0) sPlan[1] = 1;
        }
        if (tid >= 256 && tid < 256 + kSwMaxLay) {
            const int l = g + G * (tid - 256);
            if (l < fc.L) {
                int hist[kMaxSuperpR + 2], effRad[kMaxSuperpR + 2];
#pragma unroll
                for (int i = 0; i < kMaxSuperpR + 2; ++i) hist[i] = layers[l].hist[i];
                batchRadii(hist, effRad);
#pragma unroll
                for (int i = 0; i < kMaxSuperpR + 2; ++i) sEffRad[(tid - 256) * (kMaxSuperpR + 2) + i] = effRad[i];
            }
        }
        __syncthreads();
    }
    auto groupPassive = [&](int g2) { return SELF ? sPlan[2 + g2] : st->swGroupPassive[g2]; };
    if (st->errorFlags || (SELF ? sPlan[1] != 0 : st->uniformField != 0)) return;   // radius overflow / water field
    const int first = st->beamFirstInside, calcPassive = SELF ? sPlan[0] : st->firstCalculatedPassive;
    if (k < first || k >= calcPassive) return;
    // the rectangle of rays that carry dose anywhere in the field, cut into 64 x 64 patches from its own corner
    int ux0 = st->actUnion[0], uy0 = st->actUnion[1], ux1 = -st->actUnion[2], uy1 = -st->actUnion[3];
    if (ux0 > ux1 || uy0 > uy1 || ux0 < 0 || uy0 < 0) { ux0 = 0; uy0 = 0; ux1 = 0; uy1 = 0; }   // no dose at all: one patch of zeros writes the slices
    const int nPX = (ux1 - ux0) / kSwPatch + 1, nPY = (uy1 - uy0) / kSwPatchRows + 1;
    const int ppx = p % nPXg, ppy = p / nPXg;
    if (ppx >= nPX || ppy >= nPY) return;
    if (k >= groupPassive(g)) return;                                // no layer of this group deposits at k
    const int sx0 = ux0 + kSwPatch * ppx, sy0 = uy0 + kSwPatchRows * ppy;
    const int nRows = min(kSwPatchRows, uy1 - sy0 + 1), nCols = min(kSwPatch, ux1 - sx0 + 1);
    const int W = fc.W, H = fc.H, S = fc.S;
    const int nTiles = fc.tilesX * fc.tilesY;
    const int tx0 = sx0 >> 5, ty0 = sy0 >> 3;

    // ---- the group's layers that deposit at k, ascending; their dose-carrying rows inside the patch ----
    if (wv == 0) {
        const int l = g + G * lane;
        const bool on = l < fc.L && k < (SELF ? (int)sLfp[min(l, fc.L - 1)] : layers[l].layerFirstPassive);
        const unsigned long long mask = __ballot(on);
        if (on) {
            const int j = __popcll(mask & ((1ull << lane) - 1ull));
            sLay[j] = l;
            const int* act = active + ((size_t)l * S + k) * 4;
            const int ax0 = act[0], ay0 = act[1], ax1 = -act[2], ay1 = -act[3];
            int lo = max(ay0 - sy0, 0), hi = min(ay1 - sy0, nRows - 1);
            if (ax0 > sx0 + nCols - 1 || ax1 < sx0 || ax0 > ax1) { lo = 1; hi = 0; }      // no dose in this patch's columns
            if (hi < lo) { lo = 255; hi = 0; }
            sRows[j] = lo | (hi << 8);
        }
        if (lane == 0) sMisc[0] = __popcll(mask);
        if (lane < kSwNCB / kSwTk) sTicket[lane] = 0;
    }
    if (SELF) {
        __syncthreads();
        const int nL = sMisc[0];
        for (int i = tid; i < nL * kSwTileRows * kSwTileCols; i += 64 * kSwWaves) {
            const int j = i / (kSwTileRows * kSwTileCols), c = i % (kSwTileRows * kSwTileCols);
            const int ty = ty0 + c / kSwTileCols, tx = tx0 + c % kSwTileCols;
            int r = -1;
            if (ty < fc.tilesY && tx < fc.tilesX) {
                const int l = sLay[j];
                const int own = tileRad[((size_t)l * S + k) * nTiles + ty * fc.tilesX + tx];
                if (own <= kMaxSuperpR) r = sEffRad[((l - g) / G) * (kMaxSuperpR + 2) + own];
                if (r > kSwMaxR) r = -1;                             // the second launch's
            }
            sEff[i] = (signed char)r;
        }
        __syncthreads();                                             // the tables are consumed: the tile may be zeroed
    }
    for (int i = tid; i < kSwOutRows * kSwPitch; i += 64 * kSwWaves) sOut[i] = 0.0f;
    float* tab = sw + kSwLdsTab + wv * kSwWaveLds;
    for (int i = lane; i < kSwWaveLds; i += 64) tab[i] = 0.0f;       // guards (and everything the lookups may reach before it is written)
    __syncthreads();
    const int nLay = sMisc[0];
    for (int i = tid; !SELF && i < nLay * kSwTileRows * kSwTileCols; i += 64 * kSwWaves) {
        const int j = i / (kSwTileRows * kSwTileCols), c = i % (kSwTileRows * kSwTileCols);
        const int ty = ty0 + c / kSwTileCols, tx = tx0 + c % kSwTileCols;
        int r = -1;
        if (ty < fc.tilesY && tx < fc.tilesX) {
            const int l = sLay[j];
            const int own = tileRad[((size_t)l * S + k) * nTiles + ty * fc.tilesX + tx];
            if (own <= kMaxSuperpR) r = layers[l].effRad[own];       // (0xFF: not classified; overflow is reported through errorFlags)
            if (r > kSwMaxR) r = -1;                                 // the second launch's
        }
        sEff[i] = (signed char)r;
    }
    __syncthreads();

    if (dbg) dbgT1 = (long long)__builtin_amdgcn_s_memtime();
    // ---- per-lane operand addresses (float indices into the dynamic LDS), fixed for the whole block ----
    const int li = lane & 15, kq = lane >> 4;                        // MFMA 16x16x4: A[i = li][k = kq], B[k = kq][j = li], D[i = 4 kq + r][j = li]
    const int tabBase = kSwLdsTab + wv * kSwWaveLds + kq * kSwTS;    // the lane's source of quad 0
    const int idxA = tabBase + li;                                   // row |dy| = li of T
    SwIdx idxB;                                                      // table entry min(|output column - source column|, guard) per pair offset
    idxB.base = tabBase; idxB.lk = li - kq;
#pragma unroll
    for (int d = 0; d < 4; ++d) idxB.d0[d] = idxB.at(-12 + 4 * d);
    const int sx = sx0 + lane;
    const bool colOk = lane < nCols;
    const bool qTail = nCols > 4 * kSwQCut;                          // (block-uniform)
    const int effCol = (sx >> 5) - tx0;

    // ---- the wave's row-layers in (row, layer) order; dose and 1/sigma of the next one are requested before this one is multiplied ----
    struct It { int ri, j; };
    auto rowsOf = [&](int j) { return __builtin_amdgcn_readfirstlane(sRows[j]); };
    auto advance = [&](It it) {                                      // the next (row, layer slot) after `it` whose layer carries dose in that row
        for (;;) {
            if (++it.j >= nLay) { it.j = 0; it.ri += kSwWaves; }
            if (it.ri >= nRows) return it;
            const int rw = rowsOf(it.j);
            if (it.ri >= (rw & 255) && it.ri <= (rw >> 8)) return it;
        }
    };
    float doseN = 0.0f, rsN = 0.0f;
    auto request = [&](It it) {
        doseN = 0.0f; rsN = 0.0f;
        if (colOk) {
            const int l = __builtin_amdgcn_readfirstlane(sLay[it.j]);
            const size_t off = (((size_t)l * S + k) * H + (sy0 + it.ri)) * W + sx;
            doseN = bevIdd[off];
            rsN = bevRSigmaEff[off];
        }
    };
    It cur = advance(It{wv, -1});
    if (cur.ri < nRows) request(cur);
    long long dbgWait = 0, dbgFlush = 0;
    int prevRho = 0;
    for (int ri = wv; ri < nRows; ri += kSwWaves) {
        f32x4 acc[kSwNCB];
        float t16[2] = {0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < kSwNCB; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        int rhoFlush = -1;
        while (cur.ri == ri) {
            float dose = doseN;
            const float rs = rsN;
            const int j = cur.j;
            cur = advance(cur);
            if (cur.ri < nRows) request(cur);
            int rhoS = -1;
            if (dose > 0.0f) rhoS = sEff[(j * kSwTileRows + (((sy0 + ri) >> 3) - ty0)) * kSwTileCols + effCol];
            if (rhoS < 0) dose = 0.0f;
            if (!__any(rhoS >= 0)) continue;                         // the row carries no dose in this layer: exact zeros
            const int rhoRow = waveMaxI(rhoS);
            swBuild(tab + lane * kSwTS, rs, __builtin_amdgcn_sqrtf(dose), rhoS, rhoRow, prevRho);
            prevRho = rhoRow;
            rhoFlush = max(rhoFlush, rhoRow);
            swLevelRun<0>(acc, sw, idxA, idxB, qTail);
            if (rhoRow >= 1) {
                swLevelRun<1>(acc, sw, idxA, idxB, qTail);
                if (rhoRow >= 5) {
                    swLevelRun<5>(acc, sw, idxA, idxB, qTail);
                    if (rhoRow >= 9) {
                        swLevelRun<9>(acc, sw, idxA, idxB, qTail);
                        if (rhoRow >= 13) swLevelRun<13>(acc, sw, idxA, idxB, qTail);
                    }
                }
            }
            if (rhoRow == kSwMaxR) {
                // The 17th row of T (|dy| = 16; radius 16 only, ~1 % of the sources): one value per output column, on the vector ALUs:
                // t16[c] = sum_s m_s[16] m_s[|c - 16 - s|] over the sources within 16 columns of output column c (lane: c, c + 64).
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int c = lane + 64 * hf;
                    float v = 0.0f;
                    for (int dx = -kSwMaxR; dx <= kSwMaxR; ++dx) {
                        const int sI = c - kSwMaxR - dx;             // source index in the row
                        if (c < kSwOut && sI >= 0 && sI < kSwPatch) v += tab[sI * kSwTS + kSwMaxR] * tab[sI * kSwTS + abs(dx)];
                    }
                    t16[hf] += v;
                }
            }
        }
        // ---- flush T into the patch's output tile, in ascending source-row order (every element: a fixed order of additions) ----
        const long long dw0 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
        // (a ticket per group of kSwTk column blocks: row ri + 1 adds its first group while row ri adds its second)
        while (__hip_atomic_load(&sTicket[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ri) {}
        const long long dw1 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
        __builtin_amdgcn_s_setprio(3);                               // the flush is the block's serial chain: its instructions go first
        // T[|dy| = 4 kq + r][column 16 t + li] -> rows (ri + 16) +- |dy| of the tile; the minus side skips dy = 0. Plain
        // read-modify-write: the tickets make this wave the only writer of a group.
        float* outP = sOut + (ri + kSwMaxR + 4 * kq) * kSwPitch + li;
        float* outM = sOut + (ri + kSwMaxR - 4 * kq - 3) * kSwPitch + li;
#pragma unroll
        for (int t0 = 0; t0 < kSwNCB; t0 += kSwTk) {                 // kSwTk column blocks at a time: their reads in flight together, then their writes
            if (t0 > 0) while (__hip_atomic_load(&sTicket[t0 / kSwTk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ri) {}
            if (rhoFlush >= 0 && 4 * kq <= rhoFlush) {
                float oP[kSwTk][4], oM[kSwTk][4];
#pragma unroll
                for (int t = 0; t < kSwTk; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { oP[t][r] = outP[r * kSwPitch + 16 * (t0 + t)]; oM[t][r] = outM[(3 - r) * kSwPitch + 16 * (t0 + t)]; }
#pragma unroll
                for (int t = 0; t < kSwTk; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        outP[r * kSwPitch + 16 * (t0 + t)] = oP[t][r] + acc[t0 + t][r];
                        if (r > 0 || kq > 0) outM[(3 - r) * kSwPitch + 16 * (t0 + t)] = oM[t][r] + acc[t0 + t][r];
                    }
            }
            if (rhoFlush == kSwMaxR) {                               // the row |dy| = 16: t16[h2] is column 64 h2 + lane; this group's columns only
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int c = lane + 64 * h2;
                    if (c >= 16 * t0 && c < 16 * (t0 + kSwTk)) {
                        sOut[(ri + 2 * kSwMaxR) * kSwPitch + c] += t16[h2];
                        sOut[ri * kSwPitch + c] += t16[h2];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the tile writes precede the ticket
            if (lane == 0) __hip_atomic_store(&sTicket[t0 / kSwTk], ri + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_s_setprio(0);
        if (dbg) { dbgWait += dw1 - dw0; dbgFlush += (long long)__builtin_amdgcn_s_memtime() - dw1; }
    }
    const long long dbgWaveEnd = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    __syncthreads();
    if (dbg) dbgT2 = (long long)__builtin_amdgcn_s_memtime();

    // ---- hand the tile to the step's last block: agent-scope stores, drained, then the arrival counter (at the memory side) ----
    const size_t slot = (((size_t)k * nPg + p) * G + g) * kSwSlot;
    for (int i = tid; i < kSwSlot; i += 64 * kSwWaves)
        __hip_atomic_store(slots + slot + i, sOut[(i / kSwOut) * kSwPitch + i % kSwOut], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
    __syncthreads();
    int nLiveG = 0;
    for (int g2 = 0; g2 < G; ++g2) nLiveG += k < groupPassive(g2) ? 1 : 0;
    const int expected = nLiveG * nPX * nPY;
    if (tid == 0) {
        int last = 1;
        if (expected > 1) {
            const int old = __hip_atomic_fetch_add(counters + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old == expected - 1 ? 1 : 0;
            if (last) __hip_atomic_store(counters + k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        sMisc[2] = last;
    }
    __syncthreads();
    if (dbg) dbgT3 = (long long)__builtin_amdgcn_s_memtime();
    const bool lastBlock = sMisc[2] != 0;
    if (lastBlock) {
        // ---- the last block of step k: BEV slice k = sum of the tiles in (patch, group) order; zeros where no tile reaches.
        //      Four pixels x all groups of a patch are requested together (a dependent chain of loads from the memory side would cost
        //      a round trip each). ----
        const int bevW = fc.bevW, bevH = fc.bevH, nPix = bevW * bevH;
        float* out = bevDose + (size_t)k * bevW * bevH;
        // the rectangle of the slice that the patches' tiles cover (padded BEV pixel = ray + 32; tile pixel 0 = ray sx0 - 16); zeros outside
        const int bx0 = ux0 + kMaxSuperpR - kSwMaxR, by0 = uy0 + kMaxSuperpR - kSwMaxR;
        const int bx1 = min(bx0 + kSwPatch * (nPX - 1) + kSwOut, bevW), by1 = min(by0 + kSwPatchRows * (nPY - 1) + kSwOutRows, bevH);   // exclusive
        const int bw = bx1 - bx0, nIn = bw * (by1 - by0);
        for (int pix = tid; pix < nPix; pix += 64 * kSwWaves) {
            const int py = pix / bevW, px = pix - py * bevW;
            if (px < bx0 || px >= bx1 || py < by0 || py >= by1) out[pix] = 0.0f;
        }
        if (nPX == 1 && nPY == 1) {
            // One patch (the usual field): the tile rows of the groups as float4 — a quarter of the loads — through the L2 after an
            // invalidate (the other blocks' tile stores went to the memory side: sc1).
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float* sl = slots + ((size_t)k * nPg + (ppy * nPXg + ppx)) * G * kSwSlot;
            constexpr int kQ = kSwOut / 4;                           // float4 per tile row
            const int nRowsIn = by1 - by0;
            for (int i0 = tid; i0 < nRowsIn * kQ; i0 += 2 * 64 * kSwWaves) {
                float4 sum[2];
                int idx[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { sum[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); idx[u] = i0 + u * 64 * kSwWaves; if (idx[u] >= nRowsIn * kQ) idx[u] = -1; }
                for (int g0 = 0; g0 < G; g0 += 4) {
                    float4 v[4][2];
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const bool on = g0 + gg < G && k < groupPassive(min(g0 + gg, G - 1)) && idx[u] >= 0;
                            v[gg][u] = on ? *reinterpret_cast<const float4*>(sl + (size_t)(g0 + gg) * kSwSlot + 4 * idx[u]) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        }
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                        for (int u = 0; u < 2; ++u) { sum[u].x += v[gg][u].x; sum[u].y += v[gg][u].y; sum[u].z += v[gg][u].z; sum[u].w += v[gg][u].w; }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (idx[u] >= 0) {
                        const int ry = idx[u] / kQ, rx = 4 * (idx[u] - ry * kQ);
                        float* o = out + (size_t)(by0 + ry) * bevW + bx0 + rx;
                        if (bx0 + rx + 0 < bx1) o[0] = sum[u].x;
                        if (bx0 + rx + 1 < bx1) o[1] = sum[u].y;
                        if (bx0 + rx + 2 < bx1) o[2] = sum[u].z;
                        if (bx0 + rx + 3 < bx1) o[3] = sum[u].w;
                    }
            }
        } else {
        constexpr int kU = 8;
        for (int i0 = tid; i0 < nIn; i0 += kU * 64 * kSwWaves) {
            float sum[kU];
            int pixOf[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                sum[u] = 0.0f;
                const int i = i0 + u * 64 * kSwWaves;
                const int ry = i / bw, rx = i - ry * bw;
                pixOf[u] = i < nIn ? (by0 + ry) * bevW + bx0 + rx : -1;
            }
            for (int qy = 0; qy < nPY; ++qy)
                for (int qx = 0; qx < nPX; ++qx) {
                    const float* sl = slots + (((size_t)k * nPg + (qy * nPXg + qx)) * G) * kSwSlot;
                    int off[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int py = pixOf[u] / bevW, px = pixOf[u] - py * bevW;
                        const int orow = py - (by0 + kSwPatchRows * qy), ocol = px - (bx0 + kSwPatch * qx);
                        off[u] = (pixOf[u] >= 0 && (unsigned)orow < (unsigned)kSwOutRows && (unsigned)ocol < (unsigned)kSwOut) ? orow * kSwOut + ocol : -1;
                    }
                    for (int g0 = 0; g0 < G; g0 += 4) {
                        float v[4][kU];
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                            for (int u = 0; u < kU; ++u) {
                                const bool on = g0 + gg < G && k < groupPassive(min(g0 + gg, G - 1)) && off[u] >= 0;
                                v[gg][u] = on ? __hip_atomic_load(sl + (size_t)(g0 + gg) * kSwSlot + off[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
                            }
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                            for (int u = 0; u < kU; ++u) sum[u] += v[gg][u];      // (adding the 0 of a group that is not there changes nothing)
                    }
                }
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (pixOf[u] >= 0) out[pixOf[u]] = sum[u];
        }
        }
    }
    if (dbg && lane == 0) {
        long long* q2 = dbg + 8 * (size_t)gridDim.x + (4 * 16) * (size_t)blockIdx.x + 4 * wv;
        q2[0] = dbgWait; q2[1] = dbgFlush; q2[2] = dbgWaveEnd - dbgT1; q2[3] = 0;
    }
    if (dbg && tid == 0) {
        long long* q = dbg + 8 * (size_t)blockIdx.x;
        q[0] = dbgT0; q[1] = dbgT1; q[2] = dbgT2; q[3] = dbgT3; q[4] = (long long)__builtin_amdgcn_s_memtime();
        q[5] = ((long long)k << 32) | (long long)(g << 8) | (lastBlock ? 1 : 0);
        q[6] = nLay;
        q[7] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // XCC_ID | HW_ID
    }
}

}  // namespace rtd
