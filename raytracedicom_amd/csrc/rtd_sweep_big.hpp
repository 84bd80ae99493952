// K7b: the second launch of the row sweep — the sources whose batch radius is beyond the reach of k_superpose_sweep (17 .. 32; reference
// limit kernel_wrapper.cuh:432-489 / kernel_wrapper.cu:965: 32). In the fields that have such sources at all they are a few per cent of
// the sources — the deepest steps of the upper layers (measured on the bench phantom: 0.8 % at 180 deg, 1.5 % at 90 deg, 3.9 % at
// 135 deg; none at 0 / 45 / 225 deg) — but a field that had ONE used to fall back to k_superpose_mfma altogether (0.48 - 0.55 ms against
// 0.31 ms). Same formulation as rtd_sweep.hpp, sized for radius 32: tables of 33 entries + guard, a 128 x 128 output tile per 64 x 64
// patch, T in two 16-row MFMA tiles (|dy| 0..15 and 16..31) + the row |dy| = 32 on the vector ALUs, nine radius levels. One block per CU
// (139 KB of LDS), which is enough for this share of the work. A block (step k, patch, layer group) exists only at the steps where
// k_ks_plan found such a source in the group (FieldState::swBigFirst / swBigPassive, from the step range per radius class that k_fill
// records); its waves only look at the row-layers whose classification tiles hold one. The tiles are added to the BEV slice that the first
// launch has written, by the last block of the step to arrive, in (patch, group) order: bitwise reproducible like the first launch.
// (Tried first and dropped: k_superpose_mfma restricted to these sources behind the sweep — output-stationary items over sparse sources:
//  130 - 300 us for the 1 - 4 %, whichever way its items were cut.)
#pragma once
#include "rtd_sweep.hpp"

namespace rtd {

constexpr int kBgR = kMaxSuperpR;                    // 32
constexpr int kBgOut = kSwPatch + 2 * kBgR;          // 128 columns of the output tile
constexpr int kBgOutRows = kSwPatchRows + 2 * kBgR;  // 128 rows
constexpr int kBgNCB = kBgOut / 16;                  // 8 column blocks
constexpr int kBgT0 = kBgR / 16;                     // the column block that holds the patch's first source
constexpr int kBgGuard = kBgR + 1;                   // table entry 33 is always zero
constexpr int kBgTS = 35;                            // floats per source table (odd: conflict-free stores)
constexpr int kBgPitch = 132;                        // 4 rows = 528 floats = 16 banks on, like kSwPitch
constexpr int kBgSlot = kBgOutRows * kBgOut;
constexpr int kBgLdsTab = kBgOutRows * kBgPitch;
constexpr int kBgWaveLds = 64 * kBgTS;
constexpr int kBgLdsWords = kBgLdsTab + kSwWaves * kBgWaveLds;
constexpr int kBgTk = 1;                             // column blocks per flush ticket (cycles per block: 1: 116 k, 2: 118 k, 4: 144 k)
constexpr int kBgMaxGroups = 4;                      // layer groups of this launch (its partial tiles are 64 KB each)

__host__ __device__ constexpr int bgDelta(int t, int q) { return 16 * (t - kBgT0) - 4 * q; }
template <int NEED>
struct BgLevel {
    static constexpr int count() { int n = 0; for (int q = 0; q < 16; ++q) for (int t = 0; t < kBgNCB; ++t) n += swNeed(bgDelta(t, q)) == NEED ? 1 : 0; return n; }
    static constexpr int kN = count();
};
struct BgIdx {
    int base, lk;
    int d0[4];                                                       // delta = -12, -8, -4, 0
    __device__ inline int at(int delta) const { return base + min(abs(delta + lk), kBgGuard); }
};
// quads [Q0, Q1) of a level: the operands in one burst, then the MFMAs of both row tiles (|dy| = li and |dy| = 16 + li)
template <int NEED, int Q0, int Q1>
__device__ inline void bgLevelRunQ(f32x4 (&acc0)[kBgNCB], f32x4 (&acc1)[kBgNCB], const float* __restrict__ lds, int idxA, const BgIdx& ix) {
    float a0[16], a1[16], b[BgLevel<NEED>::kN];
    const int lo = NEED == 0 ? 0 : ix.at(-15 - NEED), hi = NEED == 0 ? 0 : ix.at(3 + NEED);
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        bool any = false;
#pragma unroll
        for (int t = 0; t < kBgNCB; ++t) any = any || swNeed(bgDelta(t, q)) == NEED;
        if (any) { a0[q] = lds[idxA + q * 4 * kBgTS]; a1[q] = lds[idxA + 16 + q * 4 * kBgTS]; }
    }
    {
        int n = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int t = 0; t < kBgNCB; ++t) {
                const int delta = bgDelta(t, q);
                if (swNeed(delta) == NEED) {
                    const int idx = NEED == 0 ? ix.d0[(delta + 12) >> 2] : (delta < 0 ? lo : hi);
                    if (q >= Q0 && q < Q1) b[n] = lds[idx + q * 4 * kBgTS];
                    ++n;
                }
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        int n = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int t = 0; t < kBgNCB; ++t)
                if (swNeed(bgDelta(t, q)) == NEED) {
                    if (q >= Q0 && q < Q1) {
                        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b[n], acc0[t], 0, 0, 0);
                        acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b[n], acc1[t], 0, 0, 0);
                    }
                    ++n;
                }
    }
}
// All levels of a row-layer of radius rhoRow (>= 17) for the quads [Q0, Q1). Every source of this launch has a radius of at least 17, so
// the levels 0 .. 17 always run: their operands — A once per quad, B per pair — are requested in ONE burst, then all their MFMAs follow
// (level by level this was ~24 bursts of 4 - 8 MFMAs per row-layer, each waiting for its own LDS round trip with two waves per SIMD
// to hide it: 12 k cycles per row-layer against 3.6 k of matrix work). The levels 21, 25, 29 follow one by one when the radius asks.
template <int Q0, int Q1>
struct BgBase { static constexpr int count() { int n = 0; for (int q = Q0; q < Q1; ++q) for (int t = 0; t < kBgNCB; ++t) n += swNeed(bgDelta(t, q)) <= 17 ? 1 : 0; return n; }
                static constexpr int kN = count(); };
template <int Q0, int Q1>
__device__ inline void bgRunQuads(f32x4 (&acc0)[kBgNCB], f32x4 (&acc1)[kBgNCB], const float* __restrict__ lds, int idxA, const BgIdx& ix, int rhoRow) {
    {
        float a0[16], a1[16], b[BgBase<Q0, Q1>::kN];
#pragma unroll
        for (int q = Q0; q < Q1; ++q) { a0[q] = lds[idxA + q * 4 * kBgTS]; a1[q] = lds[idxA + 16 + q * 4 * kBgTS]; }
        int off[10];                                                 // the two offsets of the levels 1, 5, 9, 13, 17
#pragma unroll
        for (int l = 0; l < 5; ++l) { off[2 * l] = ix.at(-15 - (1 + 4 * l)); off[2 * l + 1] = ix.at(3 + (1 + 4 * l)); }
        int n = 0;
#pragma unroll
        for (int q = Q0; q < Q1; ++q)
#pragma unroll
            for (int t = 0; t < kBgNCB; ++t) {
                const int delta = bgDelta(t, q), need = swNeed(delta);
                if (need <= 17) {
                    const int idx = need == 0 ? ix.d0[(delta + 12) >> 2] : off[2 * ((need - 1) >> 2) + (delta < 0 ? 0 : 1)];
                    b[n++] = lds[idx + q * 4 * kBgTS];
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        n = 0;
#pragma unroll
        for (int q = Q0; q < Q1; ++q)
#pragma unroll
            for (int t = 0; t < kBgNCB; ++t)
                if (swNeed(bgDelta(t, q)) <= 17) {
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b[n], acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b[n], acc1[t], 0, 0, 0);
                    ++n;
                }
    }
    if (rhoRow >= 21) {
        bgLevelRunQ<21, Q0, Q1>(acc0, acc1, lds, idxA, ix);
        if (rhoRow >= 25) {
            bgLevelRunQ<25, Q0, Q1>(acc0, acc1, lds, idxA, ix);
            if (rhoRow >= 29) bgLevelRunQ<29, Q0, Q1>(acc0, acc1, lds, idxA, ix);
        }
    }
}

__global__ __launch_bounds__(64 * kSwWaves, 2) void k_superpose_sweep_big(const float* __restrict__ bevIdd, const float* __restrict__ bevRSigmaEff,
                                                                       const unsigned char* __restrict__ tileRad, const LayerPlan* __restrict__ layers,
                                                                       const FieldState* __restrict__ st, FieldConst fc, int G, int nPXg, int nPYg,
                                                                       const int* __restrict__ active, float* __restrict__ slots, int* __restrict__ counters,
                                                                       float* __restrict__ bevDose, long long* __restrict__ dbg) {
    extern __shared__ float sw[];
    // diagnostic build only (RTD_SWEEP_DEBUG): clock stamps per block — no output value depends on them
    const long long dbgT0 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    long long dbgT1 = 0, dbgT2 = 0, dbgT3 = 0;
    int dbgBig = 0;
    long long dbgBuild = 0, dbgMul = 0, dbgFlush = 0;
    __shared__ signed char sEff[kSwMaxLay * kSwTileRows * kSwTileCols];   // [layer slot][9][3] batch radius of the tile if it is this launch's (-1: not)
    __shared__ int sRows[kSwMaxLay];
    __shared__ int sLay[kSwMaxLay];
    __shared__ int sMisc[4];
    __shared__ int sTicket[kBgNCB];                                  // per column block: the source row whose turn it is to flush
    float* sOut = sw;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int item = blockIdx.x;
    const int g = item % G; item /= G;
    const int nPg = nPXg * nPYg;
    const int k = fc.S - 1 - item % fc.S; item /= fc.S;
    const int p = item;
    if (st->errorFlags || st->uniformField || st->maxRadius <= kSwMaxR) return;   // radius overflow / water field / nothing beyond the first launch
    if (k < st->swBigFirst[g] || k >= st->swBigPassive[g]) return;   // no layer of the group has such a source at k
    int ux0 = st->actUnion[0], uy0 = st->actUnion[1], ux1 = -st->actUnion[2], uy1 = -st->actUnion[3];
    if (ux0 > ux1 || uy0 > uy1 || ux0 < 0 || uy0 < 0) return;        // (no dose at all: no such source either)
    const int nPX = (ux1 - ux0) / kSwPatch + 1, nPY = (uy1 - uy0) / kSwPatchRows + 1;      // the patches of the first launch
    const int ppx = p % nPXg, ppy = p / nPXg;
    if (ppx >= nPX || ppy >= nPY) return;
    const int sx0 = ux0 + kSwPatch * ppx, sy0 = uy0 + kSwPatchRows * ppy;
    const int nRows = min(kSwPatchRows, uy1 - sy0 + 1), nCols = min(kSwPatch, ux1 - sx0 + 1);
    const int W = fc.W, H = fc.H, S = fc.S;
    const int nTiles = fc.tilesX * fc.tilesY;
    const int tx0 = sx0 >> 5, ty0 = sy0 >> 3;

    if (wv == 0) {
        const int l = g + G * lane;
        const bool on = l < fc.L && k < layers[l].layerFirstPassive;
        const unsigned long long mask = __ballot(on);
        if (on) {
            const int j = __popcll(mask & ((1ull << lane) - 1ull));
            sLay[j] = l;
            const int* act = active + ((size_t)l * S + k) * 4;
            const int ax0 = act[0], ay0 = act[1], ax1 = -act[2], ay1 = -act[3];
            int lo = max(ay0 - sy0, 0), hi = min(ay1 - sy0, nRows - 1);
            if (ax0 > sx0 + nCols - 1 || ax1 < sx0 || ax0 > ax1) { lo = 1; hi = 0; }
            if (hi < lo) { lo = 255; hi = 0; }
            sRows[j] = lo | (hi << 8);
        }
        if (lane == 0) sMisc[0] = __popcll(mask);
        if (lane < kBgNCB) sTicket[lane] = 0;
    }
    for (int i = tid; i < kBgOutRows * kBgPitch; i += 64 * kSwWaves) sOut[i] = 0.0f;
    float* tab = sw + kBgLdsTab + wv * kBgWaveLds;
    for (int i = lane; i < kBgWaveLds; i += 64) tab[i] = 0.0f;
    __syncthreads();
    const int nLay = sMisc[0];
    for (int i = tid; i < nLay * kSwTileRows * kSwTileCols; i += 64 * kSwWaves) {
        const int j = i / (kSwTileRows * kSwTileCols), c = i % (kSwTileRows * kSwTileCols);
        const int ty = ty0 + c / kSwTileCols, tx = tx0 + c % kSwTileCols;
        int r = -1;
        if (ty < fc.tilesY && tx < fc.tilesX) {
            const int l = sLay[j];
            const int own = tileRad[((size_t)l * S + k) * nTiles + ty * fc.tilesX + tx];
            if (own <= kMaxSuperpR) r = layers[l].effRad[own];
            if (r <= kSwMaxR) r = -1;                                // the first launch's
        }
        sEff[i] = (signed char)r;
    }
    __syncthreads();

    if (dbg) dbgT1 = (long long)__builtin_amdgcn_s_memtime();
    const int li = lane & 15, kq = lane >> 4;
    const int tabBase = kBgLdsTab + wv * kBgWaveLds + kq * kBgTS;
    const int idxA = tabBase + li;
    BgIdx idxB;
    idxB.base = tabBase; idxB.lk = li - kq;
#pragma unroll
    for (int d = 0; d < 4; ++d) idxB.d0[d] = idxB.at(-12 + 4 * d);
    const int sx = sx0 + lane;
    const bool colOk = lane < nCols;
    const int effCol = (sx >> 5) - tx0;

    struct It { int ri, j; };
    auto advance = [&](It it) {       // the next (row, layer slot) whose layer carries dose in that row AND whose tiles hold a source of this launch
        for (;;) {
            if (++it.j >= nLay) { it.j = 0; it.ri += kSwWaves; }
            if (it.ri >= nRows) return it;
            const int rw = __builtin_amdgcn_readfirstlane(sRows[it.j]);
            if (it.ri < (rw & 255) || it.ri > (rw >> 8)) continue;
            const signed char* e = sEff + (it.j * kSwTileRows + (((sy0 + it.ri) >> 3) - ty0)) * kSwTileCols;
            const int any = max(max((int)e[0], (int)e[1]), (int)e[2]);
            if (__builtin_amdgcn_readfirstlane(any) >= 0) return it;
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
    int prevRho = 0;
    for (int ri = wv; ri < nRows; ri += kSwWaves) {
        f32x4 acc0[kBgNCB], acc1[kBgNCB];
        float tR[2] = {0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < kBgNCB; ++t) { acc0[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; acc1[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
        int rhoFlush = -1, qmRow = 0;
        while (cur.ri == ri) {
            float dose = doseN;
            const float rs = rsN;
            const int j = cur.j;
            cur = advance(cur);
            if (cur.ri < nRows) request(cur);
            int rhoS = -1;
            if (dose > 0.0f) rhoS = sEff[(j * kSwTileRows + (((sy0 + ri) >> 3) - ty0)) * kSwTileCols + effCol];
            if (rhoS < 0) dose = 0.0f;
            const unsigned long long liveS = __ballot(rhoS >= 0);
            if (!liveS) continue;
            const int qm = ((liveS & 0xFFFFull) ? 1 : 0) | ((liveS & 0xFFFF0000ull) ? 2 : 0) | ((liveS & 0xFFFF00000000ull) ? 4 : 0) | ((liveS >> 48) ? 8 : 0);
            const int rhoRow = waveMaxI(rhoS);                       // >= 17
            ++dbgBig;
            const long long db0 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
            swBuild(tab + lane * kBgTS, rs, __builtin_amdgcn_sqrtf(dose), rhoS, rhoRow, prevRho);
            prevRho = rhoRow;
            rhoFlush = max(rhoFlush, rhoRow);
            qmRow |= qm;
            const long long db1 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
            // per run of four quads (16 sources): only the runs that hold a source of this launch at all — the others' tables are zeros
            // (a classification tile is 32 sources wide, and often one of a row's two is this launch's)
            if (qm & 1) bgRunQuads<0, 4>(acc0, acc1, sw, idxA, idxB, rhoRow);
            if (qm & 2) bgRunQuads<4, 8>(acc0, acc1, sw, idxA, idxB, rhoRow);
            if (qm & 4) bgRunQuads<8, 12>(acc0, acc1, sw, idxA, idxB, rhoRow);
            if (qm & 8) bgRunQuads<12, 16>(acc0, acc1, sw, idxA, idxB, rhoRow);
            if (dbg) { dbgBuild += db1 - db0; dbgMul += (long long)__builtin_amdgcn_s_memtime() - db1; }
            if (rhoRow == kBgR) {
                // the 33rd row of T (|dy| = 32; radius 32 only): one value per output column on the vector ALUs
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int c = lane + 64 * hf;
                    float v = 0.0f;
                    for (int dx = -kBgR; dx <= kBgR; ++dx) {
                        const int sI = c - kBgR - dx;
                        if (sI >= 0 && sI < kSwPatch) v += tab[sI * kBgTS + kBgR] * tab[sI * kBgTS + abs(dx)];
                    }
                    tR[hf] += v;
                }
            }
        }
        // ---- flush, in ascending source-row order — per COLUMN BLOCK: eight tickets, so that the rows' flushes overlap like a pipeline
        //      (row ri adds its block t while row ri + 1 adds its block t - 1; an element still receives its addends in row order).
        //      With one ticket for the whole row the 64 flushes of a block were its critical path (measured: 3.4 k cycles each,
        //      220 k of a block's 245 k, whatever the number of sources). (Requesting block t + 1's tile values before block t is written —
        //      one LDS round trip per step instead of two — measured slower, 199 k against 134 k: a row then holds block t until it has
        //      block t + 1's ticket, and the rows move in a convoy.) ----
        const long long df0 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
        // column blocks the row's sources can reach: run i of four quads (sources 16 i ...) -> tile columns [16 i + 32 - rho, 16 i + 47 + rho];
        // the others only pass their ticket on
        int blockMask = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (qmRow & (1 << i)) {
                const int lo = max((16 * i + 32 - rhoFlush) >> 4, 0), hi = min((16 * i + 47 + rhoFlush) >> 4, kBgNCB - 1);
                blockMask |= ((2 << hi) - 1) & ~((1 << lo) - 1);
            }
        if (rhoFlush == kBgR) blockMask = (1 << kBgNCB) - 1;         // (the row |dy| = 32 is added to every block)
#pragma unroll
        for (int tg = 0; tg < kBgNCB / kBgTk; ++tg) {
            while (__hip_atomic_load(&sTicket[tg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ri) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_s_setprio(3);                           // the flush is the block's serial chain: its instructions go first
            if (rhoFlush >= 0 && ((blockMask >> (kBgTk * tg)) & ((1 << kBgTk) - 1)) != 0) {
                // T[|dy| = dyBase + 4 kq + r][column 16 t + li] -> rows (ri + 32) +- |dy| of the tile; the minus side skips dy = 0
                float* outP = sOut + (ri + kBgR + 4 * kq) * kBgPitch + 16 * kBgTk * tg + li;
                float* outM = sOut + (ri + kBgR - 4 * kq - 3) * kBgPitch + 16 * kBgTk * tg + li;
                const bool on0 = 4 * kq <= rhoFlush, on1 = 16 + 4 * kq <= rhoFlush;
                // (four predicated regions per step — the reads of the two row tiles, then their writes: with the predicate inside the
                //  per-register loops the compiler made a branch around every pair of accesses)
                float oP0[kBgTk][4], oM0[kBgTk][4], oP1[kBgTk][4], oM1[kBgTk][4];
                if (on0) {
#pragma unroll
                    for (int u = 0; u < kBgTk; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { oP0[u][r] = outP[r * kBgPitch + 16 * u]; oM0[u][r] = outM[(3 - r) * kBgPitch + 16 * u]; }
                }
                if (on1) {
#pragma unroll
                    for (int u = 0; u < kBgTk; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { oP1[u][r] = outP[(16 + r) * kBgPitch + 16 * u]; oM1[u][r] = outM[(3 - r - 16) * kBgPitch + 16 * u]; }
                }
                if (on0) {
#pragma unroll
                    for (int u = 0; u < kBgTk; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            outP[r * kBgPitch + 16 * u] = oP0[u][r] + acc0[kBgTk * tg + u][r];
                            if (r > 0 || kq > 0) outM[(3 - r) * kBgPitch + 16 * u] = oM0[u][r] + acc0[kBgTk * tg + u][r];
                        }
                }
                if (on1) {
#pragma unroll
                    for (int u = 0; u < kBgTk; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            outP[(16 + r) * kBgPitch + 16 * u] = oP1[u][r] + acc1[kBgTk * tg + u][r];
                            outM[(3 - r - 16) * kBgPitch + 16 * u] = oM1[u][r] + acc1[kBgTk * tg + u][r];
                        }
                }
                if (rhoFlush == kBgR) {
#pragma unroll
                    for (int u = 0; u < kBgTk; ++u) {
                        const int t = kBgTk * tg + u;
                        if (kq == (t & 3)) {                         // the row |dy| = 32: block t's 16 columns are held by the lanes of quarter t & 3, half t >> 2
                            const float v = (t >> 2) ? tR[1] : tR[0];
                            sOut[(ri + 2 * kBgR) * kBgPitch + 16 * t + li] += v;
                            sOut[ri * kBgPitch + 16 * t + li] += v;
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the tile writes precede the ticket
            if (lane == 0) __hip_atomic_store(&sTicket[tg], ri + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_s_setprio(0);
        }
        if (dbg) dbgFlush += (long long)__builtin_amdgcn_s_memtime() - df0;
    }
    const long long dbgWaveEnd = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    __syncthreads();
    if (dbg) dbgT2 = (long long)__builtin_amdgcn_s_memtime();

    // ---- hand the tile to the step's last block (agent-scope stores, drained, then the arrival counter at the memory side) ----
    const size_t slot = (((size_t)k * nPg + p) * G + g) * kBgSlot;
    for (int i = tid; i < kBgSlot; i += 64 * kSwWaves)
        __hip_atomic_store(slots + slot + i, sOut[(i / kBgOut) * kBgPitch + i % kBgOut], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
    __syncthreads();
    int nLiveG = 0;
    for (int g2 = 0; g2 < G; ++g2) nLiveG += (k >= st->swBigFirst[g2] && k < st->swBigPassive[g2]) ? 1 : 0;
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
    auto stamp = [&](int last) {
        if (dbg && lane == 0) {
            long long* q = dbg + 48 * (size_t)blockIdx.x;
            if (wv == 0) { q[0] = dbgT0; q[1] = dbgT1; q[2] = dbgT2; q[3] = dbgT3; q[4] = (long long)__builtin_amdgcn_s_memtime(); q[5] = ((long long)k << 32) | (long long)(g << 8) | last; q[6] = nLay; }
            q[8 + wv] = dbgBig;
            q[16 + 4 * wv] = dbgBuild; q[17 + 4 * wv] = dbgMul; q[18 + 4 * wv] = dbgFlush; q[19 + 4 * wv] = dbgWaveEnd - dbgT1;
        }
    };
    if (sMisc[2] == 0) { stamp(0); return; }
    // ---- the last block of step k adds the tiles, in (patch, group) order, to the slice the first launch has written ----
    const int bevW = fc.bevW, bevH = fc.bevH;
    float* out = bevDose + (size_t)k * bevW * bevH;
    const int bx0 = ux0 + kMaxSuperpR - kBgR, by0 = uy0 + kMaxSuperpR - kBgR;     // padded BEV pixel = ray + 32; tile pixel 0 = ray sx0 - 32
    const int bx1 = min(bx0 + kSwPatch * (nPX - 1) + kBgOut, bevW), by1 = min(by0 + kSwPatchRows * (nPY - 1) + kBgOutRows, bevH);   // exclusive
    // Tile by tile, as float4 through the L2 after an invalidate (the other blocks' tile stores went to the memory side: sc1); all
    // groups of four pixel quads are requested together — a chain of dependent loads from the memory side costs a round trip each.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    constexpr int kQ = kBgOut / 4;                                   // float4 per tile row
    constexpr int kU = 4;
    for (int qy = 0; qy < nPY; ++qy)
        for (int qx = 0; qx < nPX; ++qx) {
            const float* sl = slots + (((size_t)k * nPg + (qy * nPXg + qx)) * G) * kBgSlot;
            const int ox = bx0 + kSwPatch * qx, oy = by0 + kSwPatchRows * qy;     // BEV pixel of the tile's pixel (0, 0)
            for (int i0 = tid; i0 < kBgOutRows * kQ; i0 += kU * 64 * kSwWaves) {
                float4 sum[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) sum[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                for (int g0 = 0; g0 < G; g0 += 2) {
                    float4 v[2][kU];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g2 = min(g0 + gg, G - 1);
                        const bool on = g0 + gg < G && k >= st->swBigFirst[g2] && k < st->swBigPassive[g2];
#pragma unroll
                        for (int u = 0; u < kU; ++u) {
                            const int idx = i0 + u * 64 * kSwWaves;
                            v[gg][u] = (on && idx < kBgOutRows * kQ) ? *reinterpret_cast<const float4*>(sl + (size_t)g2 * kBgSlot + 4 * idx) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        }
                    }
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                        for (int u = 0; u < kU; ++u) { sum[u].x += v[gg][u].x; sum[u].y += v[gg][u].y; sum[u].z += v[gg][u].z; sum[u].w += v[gg][u].w; }
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int idx = i0 + u * 64 * kSwWaves;
                    const int ry = idx / kQ, rx = 4 * (idx - ry * kQ);
                    const int py = oy + ry, px = ox + rx;
                    if (idx < kBgOutRows * kQ && py < by1 && (sum[u].x != 0.0f || sum[u].y != 0.0f || sum[u].z != 0.0f || sum[u].w != 0.0f)) {
                        float* o = out + (size_t)py * bevW + px;
                        if (px + 0 < bx1) o[0] += sum[u].x;
                        if (px + 1 < bx1) o[1] += sum[u].y;
                        if (px + 2 < bx1) o[2] += sum[u].z;
                        if (px + 3 < bx1) o[3] += sum[u].w;
                    }
                }
            }
            __syncthreads();                                         // the next patch's tile overlaps this one's pixels
        }
    stamp(1);
}

}  // namespace rtd
