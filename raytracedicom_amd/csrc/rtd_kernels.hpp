// rtd_kernels.hpp — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the pencil-beam dose path.
//
// One kernel per stage of the reference's cudaWrapperProtons (src/kernel_wrapper.cu:381-1369). There are no
// texture units on gfx950: every CT / LUT / BEV interpolation is written out against plain global or LDS
// memory with the BORDER / CLAMP semantics of the reference's samplers (kernel_wrapper.cu:418-537).
// All control scalars (entry step, cut-off steps, tile-radius histograms, work lists, bounding box) stay
// on the device, so a field is a fixed sequence of launches with no host round trip.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>

#include "rtd_geometry.hpp"

namespace rtd {

constexpr int kWave = 64;
constexpr int kKsTileX = 64, kKsTileY = 32;   // superposition: output tile owned by one wave (4 x 2 MFMA tiles)
constexpr int kKsMaxOrder = 64;               // superposition: output tiles ranked by expected work when there are at most this many
constexpr int kMaxLayers = 256;
constexpr int kMaxSteps = 4096;
constexpr int kNoRadius = 0xFF;

enum FieldError : int { kErrRadiusOverflow = 1, kErrPackOverflow = 2 };

struct LutView {
    const float* density; int nDensity;
    const float* sp; int nSp;
    const float* rrl; int nRrl;
    const float* cidd; int nSamples; int nEnergies;
    const float* nucWeight; const float* nucSqSigma;   // NUCLEAR_CORR tables, [nEnergies][nSamples] like cidd (null when absent)
};

// Per-layer record. Host fills the beam-model part at field creation; k_plan / k_fill / k_ks_plan fill the rest.
struct LayerPlan {
    float energyIdx, energyScaleFact, peakDepth;       // kernel_wrapper.cu:834-837
    float spotSigmaX, spotSigmaY;                      // BeamSettings::getSpotSigmas
    float airCoefA, airCoefB;                          // sigmaSqAirCoefs(peakDepth)  fill_idd_and_sigma_params.cu:74-83
    float sigmaSqAirLin, sigmaSqAirQuad;               // initStepAndAirDiv           fill_idd_and_sigma_params.cu:28-40
    float entrySigmaX, entrySigmaY;                    // kernel_wrapper.cu:838-841   (device)
    int afterLast;                                     // kernel_wrapper.cu:923-924   (device)
    int layerFirstPassive;                             // kernel_wrapper.cu:952-957   (device, atomicMax)
    int hist[kMaxSuperpR + 2];                         // tilePrimRadCtrs             kernel_wrapper.cu:959-963
    int effRad[kMaxSuperpR + 2];                       // batch radius per tile radius kernel_wrapper.cu:966-976
    int classLo[kMaxSuperpR + 2], classHi[kMaxSuperpR + 2];   // first / last step at which a tile of the layer has that radius class (k_fill)
};

struct FieldState {
    int beamFirstInside;            // kernel_wrapper.cu:781-784
    int beamFirstOutside;           // :785-787
    int firstGuaranteedPassive;     // :796
    int firstCalculatedPassive;     // :955-957
    float entryZ, pxSpMultX, pxSpMultY;   // :784, :849
    int errorFlags;
    int maxRadius;
    long long liveSteps;
    int bboxMin[3], bboxMax[3];     // :1207-1208
    int tboxMin[3], tboxMax[3];     // sub-box of it that can receive dose: image of the non-zero BEV rectangle (transfer loops over this)
    TransferParams transfer;        // :1213
    int empty;                      // nothing inside the patient for this beam
    int groupPassive[32];           // per superposition layer group: first step at which none of its layers deposits
    int swGroupPassive[16];         // the same for the layer groups of k_superpose_sweep (rtd_sweep.hpp: its own, smaller group count)
    // ... and, per layer group of the sweep's SECOND launch (rtd_sweep_big.hpp), the steps [swBigFirst, swBigPassive) at which some
    // layer of the group has a tile whose batch radius is beyond the reach of the first launch (16)
    int swBigFirst[16], swBigPassive[16];
    int actUnion[4];                // minima of (x, y, -x, -y) over all rays that carry dose in any (layer, step)
    int bevLo[2], bevHi[2];         // padded-BEV rectangle outside which every slice is exactly zero (transfer early-out)
    // The slab the transfer samples: packW x packH pixels per slice, pixel (0, 0) = padded-BEV pixel (packX0, packY0), first
    // slice = slice slabFirst of the buffer. The field's own BEV buffer: (0, 0, bevW, bevH, beamFirstInside); a slab exported
    // by k_pack_bev for another GPU: the rectangle that carries dose, slices from 0.
    int packX0, packY0, packW, packH, slabFirst;
    // Uniform-sigma fields (water): k_fill raises nonUniform when the live rays of a (layer, step, tile) differ in sigma^2; k_ks_plan
    // sets uniformField when no tile did and every depositing (layer, step) slice has ONE sigma^2 over all its tiles — the
    // superposition of such a slice is a separable convolution (rtd_uniform.hpp) and the general superposition stands aside.
    int nonUniform, uniformField;
    unsigned short fillItems[2 * 256];      // (layer << 1 | role) of k_fill's walks by descending cost (k_plan), for its block placement
    unsigned char tileOrder[kKsMaxOrder];   // superposition dispatch order of the output tiles: most source rays in reach first
};

// Host-known per-field constants, passed by value.
struct FieldConst {
    int W, H, L, S;                 // ray grid (primRayDims) and tracer steps
    int bevW, bevH;                 // W+64, H+64
    int tilesX, tilesY;
    float rayRes[3], rayOffset[3];
    float sourceDist[2];
    int spotNx, spotNy;
    float spotDelta[3], spotOffset[3];
    float maxPeakDepth;             // kernel_wrapper.cu:792-794
    float bpDepthCutoff, convSigmaCutoff, ksSigmaCutoff, rayWeightCutoff;
    int doseToWater, nozzle;
    // NUCLEAR_CORR (default off; include/rtd.h: RTD_NUC_*): variant, nuclear grid = spot grid rounded up to whole tiles
    // (kernel_wrapper.cu:667), spot pitch in rays (:922)
    int nuclearCorr, nucW, nucH;
    float spotDist;
};

// ------------------------------------------------------------------------------------------------
// wave64 helpers: butterfly reductions on DPP (quad_perm, row_half_mirror, row_mirror, row_bcast15/31) — VALU only, no
// LDS crossbar (ds_bpermute) round trips; the total lands in lane 63 and is broadcast with v_readlane.
template <typename T, typename Op>
__device__ inline T waveReduce(T v, Op op) {
    int x = __builtin_bit_cast(int, v);
#define RTD_DPP_STEP(ctrl, rmask) { int t = __builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xF, false); \
                                    x = __builtin_bit_cast(int, op(__builtin_bit_cast(T, x), __builtin_bit_cast(T, t))); }
    RTD_DPP_STEP(0xB1, 0xF)    // quad_perm [1,0,3,2]
    RTD_DPP_STEP(0x4E, 0xF)    // quad_perm [2,3,0,1]
    RTD_DPP_STEP(0x141, 0xF)   // row_half_mirror
    RTD_DPP_STEP(0x140, 0xF)   // row_mirror     -> every lane of a 16-lane row holds the row's result
    RTD_DPP_STEP(0x142, 0xA)   // row_bcast:15 into rows 1 and 3
    RTD_DPP_STEP(0x143, 0xC)   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's result
#undef RTD_DPP_STEP
    return __builtin_bit_cast(T, __builtin_amdgcn_readlane(x, 63));
}
__device__ inline float waveMin(float v) { return waveReduce(v, [](float a, float b) { return b < a ? b : a; }); }
// minimum over each 32-lane half of the wave; valid in lanes 31 and 63 (quad_perm, row_half_mirror, row_mirror, row_bcast15)
__device__ inline float halfWaveMin(float v) {
    int x = __builtin_bit_cast(int, v);
#define RTD_MIN_STEP(ctrl, rmask) { int t = __builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xF, false); \
                                    const float a = __builtin_bit_cast(float, x), b = __builtin_bit_cast(float, t); x = __builtin_bit_cast(int, b < a ? b : a); }
    RTD_MIN_STEP(0xB1, 0xF)    // quad_perm [1,0,3,2]
    RTD_MIN_STEP(0x4E, 0xF)    // quad_perm [2,3,0,1]
    RTD_MIN_STEP(0x141, 0xF)   // row_half_mirror
    RTD_MIN_STEP(0x140, 0xF)   // row_mirror
    RTD_MIN_STEP(0x142, 0xA)   // row_bcast15 into rows 1 and 3
#undef RTD_MIN_STEP
    return __builtin_bit_cast(float, x);
}
__device__ inline int waveMinI(int v) { return waveReduce(v, [](int a, int b) { return b < a ? b : a; }); }
__device__ inline int waveMaxI(int v) { return waveReduce(v, [](int a, int b) { return b > a ? b : a; }); }
// Workgroup barrier that orders LDS only: __syncthreads() also waits for every global load in flight (its fence covers all address
// spaces: s_waitcnt vmcnt(0) in front of s_barrier), which defeats a prefetch that is meant to stay in flight across the barrier.
// For barriers that hand over LDS data only.
__device__ inline void ldsBarrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ inline int roundToI(int v, int m) { return ((v + m - 1) / m) * m; }   // roundTo, kernel_wrapper.cu:45-48
__device__ inline int f2iSat(float v) { return (int)v; }   // v_cvt_i32_f32: NaN -> 0, saturating (same as the reference GPU)

// ------------------------------------------------------------------------------------------------
// Software samplers (replace tex1D/tex2D/tex3D, kernel_wrapper.cu:418-537). p = coordinate without the +0.5.
__device__ inline float lerpW(float a, float v0, float v1) { return (1.0f - a) * v0 + a * v1; }

template <typename Ptr>
__device__ inline float sample1dClamp(Ptr t, int n, float p) {
    float fl = floorf(p);
    float a = p - fl;
    int i0 = (int)fl, i1 = i0 + 1;
    if (!(p >= 0.0f)) { i0 = 0; i1 = 0; a = 0.0f; }
    i0 = i0 > n - 1 ? n - 1 : i0;
    i1 = i1 > n - 1 ? n - 1 : i1;
    return lerpW(a, t[i0], t[i1]);
}
__device__ inline float sample2dClamp(const float* __restrict__ t, int ncol, int nrow, float px, float py) {
    float fx = floorf(px), fy = floorf(py);
    float ax = px - fx, ay = py - fy;
    int x0 = (int)fx, x1 = x0 + 1, y0 = (int)fy, y1 = y0 + 1;
    if (!(px >= 0.0f)) { x0 = 0; x1 = 0; ax = 0.0f; }
    if (!(py >= 0.0f)) { y0 = 0; y1 = 0; ay = 0.0f; }
    x0 = x0 > ncol - 1 ? ncol - 1 : x0; x1 = x1 > ncol - 1 ? ncol - 1 : x1;
    y0 = y0 > nrow - 1 ? nrow - 1 : y0; y1 = y1 > nrow - 1 ? nrow - 1 : y1;
    float r0 = lerpW(ax, t[(size_t)y0 * ncol + x0], t[(size_t)y0 * ncol + x1]);
    float r1 = lerpW(ax, t[(size_t)y1 * ncol + x0], t[(size_t)y1 * ncol + x1]);
    return lerpW(ay, r0, r1);
}
__device__ inline float fetch3dBorder(const float* __restrict__ vol, int nx, int ny, int nz, int x, int y, int z) {
    bool in = (unsigned)x < (unsigned)nx && (unsigned)y < (unsigned)ny && (unsigned)z < (unsigned)nz;
    size_t idx = in ? ((size_t)z * ny + y) * nx + x : 0;
    float v = vol[idx];
    return in ? v : 0.0f;
}
__device__ inline float sample3dBorder(const float* __restrict__ vol, int nx, int ny, int nz, float px, float py, float pz) {
    if (!(px > -1.0f && py > -1.0f && pz > -1.0f && px < (float)nx && py < (float)ny && pz < (float)nz)) return 0.0f;
    float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    float ax = px - fx, ay = py - fy, az = pz - fz;
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    float v000, v100, v010, v110, v001, v101, v011, v111;
    if (x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + 1 < nx && y0 + 1 < ny && z0 + 1 < nz) {
        // interior cell (almost every sample): one index, eight plain loads, no per-corner bounds logic
        const float* p = vol + ((size_t)(unsigned)(z0 * ny + y0) * (unsigned)nx + (unsigned)x0);
        const size_t sxy = (size_t)(unsigned)nx * (unsigned)ny;
        v000 = p[0]; v100 = p[1]; v010 = p[nx]; v110 = p[nx + 1];
        v001 = p[sxy]; v101 = p[sxy + 1]; v011 = p[sxy + nx]; v111 = p[sxy + nx + 1];
    } else {
        v000 = fetch3dBorder(vol, nx, ny, nz, x0, y0, z0);         v100 = fetch3dBorder(vol, nx, ny, nz, x0 + 1, y0, z0);
        v010 = fetch3dBorder(vol, nx, ny, nz, x0, y0 + 1, z0);     v110 = fetch3dBorder(vol, nx, ny, nz, x0 + 1, y0 + 1, z0);
        v001 = fetch3dBorder(vol, nx, ny, nz, x0, y0, z0 + 1);     v101 = fetch3dBorder(vol, nx, ny, nz, x0 + 1, y0, z0 + 1);
        v011 = fetch3dBorder(vol, nx, ny, nz, x0, y0 + 1, z0 + 1); v111 = fetch3dBorder(vol, nx, ny, nz, x0 + 1, y0 + 1, z0 + 1);
    }
    float c00 = lerpW(ax, v000, v100);
    float c10 = lerpW(ax, v010, v110);
    float c01 = lerpW(ax, v001, v101);
    float c11 = lerpW(ax, v011, v111);
    float c0 = lerpW(ay, c00, c10);
    float c1 = lerpW(ay, c01, c11);
    return lerpW(az, c0, c1);
}

// ------------------------------------------------------------------------------------------------
// K0: reset of the per-field device state (the reference re-creates these per beam, kernel_wrapper.cu:685-734). No launch of
// its own: the scalars that the tracer's scan accumulates into are reset by the first thread of the sampling kernel (the
// launch before the scan), the per-layer records and the two fills that the reference does with cudaMemset per layer
// (kernel_wrapper.cu:824-827) by the waves of the scan kernel that have no serial chain to walk.
__device__ inline void resetFieldScalars(FieldState* st) {
    st->beamFirstInside = 0x7fffffff; st->beamFirstOutside = -0x7fffffff; st->firstGuaranteedPassive = 0;
    st->firstCalculatedPassive = 0; st->errorFlags = 0; st->maxRadius = 0; st->liveSteps = 0;
    st->empty = 0; st->nonUniform = 0; st->uniformField = 0;
    for (int i = 0; i < 4; ++i) st->actUnion[i] = 0x7fffffff;
    for (int i = 0; i < 3; ++i) { st->bboxMin[i] = 0; st->bboxMax[i] = 0; st->tboxMin[i] = 0; st->tboxMax[i] = -1; }
}
struct ResetJob {
    LayerPlan* layers; int L;
    unsigned int* tileRadWords; size_t nRadWords;
    int* active; size_t nActive;
    float* nucIdd; float* nucRs; size_t nNuc;      // NUCLEAR_CORR: (0, inf) = the reference's fills at kernel_wrapper.cu:862-863
    unsigned int* sigMin; unsigned int* sigMax; size_t nSig;   // per (layer, step): bits of the smallest / largest tile-uniform sigma^2
    long long* scanDbg;             // diagnostic build only (RTD_SCAN_DEBUG): clock stamps of k_trace_scan's blocks, 8 per block
};
__device__ inline void resetFieldArrays(const ResetJob& j, size_t t, size_t nT) {
    for (size_t l = t; l < (size_t)j.L; l += nT) {
        j.layers[l].layerFirstPassive = 0; j.layers[l].afterLast = 0;
        for (int i = 0; i < kMaxSuperpR + 2; ++i) { j.layers[l].hist[i] = 0; j.layers[l].effRad[i] = i; j.layers[l].classLo[i] = 0x7fffffff; j.layers[l].classHi[i] = -1; }
    }
    for (size_t i = t; i < j.nRadWords; i += nT) j.tileRadWords[i] = 0xFFFFFFFFu;    // every (layer, step, tile): "not classified"
    for (size_t i = t; i < j.nActive; i += nT) j.active[i] = 0x7f7f7f7f;             // empty dose rectangles (+large minima)
    for (size_t i = t; i < j.nNuc; i += nT) { j.nucIdd[i] = 0.0f; j.nucRs[i] = __int_as_float(0x7f800000); }
    for (size_t i = t; i < j.nSig; i += nT) { j.sigMin[i] = 0x7f800000u; j.sigMax[i] = 0u; }
}

// ------------------------------------------------------------------------------------------------
// K1: ray tracer = fillBevDensityAndSp (kernel_wrapper.cu:130-187), split in two passes so that the 8 scattered CT
// loads per sample (no texture unit on gfx950) run with (rays x segments) parallelism instead of one serial
// 512-step walk per ray:
//   k_trace_sample  one thread per (ray, segment of kTraceSeg steps): trilinear HU sample, density LUT, and the
//                   step's stopping-power term stepLen*SP(hu). The sample position is advanced with the
//                   reference's repeated `pos += step` (arithmetic only) so it is the same float sequence.
//   k_trace_scan    three waves per 64 rays: the reference's sequential sums (cumulSp, cumulHuPlus1000) and entry/exit
//                   logic over the stored terms, fused with the int reductions that follow the tracer in the
//                   reference (sliceMin/MaxVar<int>, kernel_wrapper.cu:781-787).
// Rays are numbered row-major; lanes hold consecutive rays, so all stores are coalesced and step-major.
constexpr int kTraceSeg = 9, kTraceSegsPerBlock = 2;   // a block = 256 rays x 2 segments (8 waves share one copy of the LUT rows in LDS)

// The sample positions are the serial `pos += step` sequence of the reference walk (kernel_wrapper.cu:183) — not start + k * step in
// floating point — so a thread that starts at step k0 has to know the k0-th term. They depend on the field's geometry only: the
// positions at the segment boundaries are walked once, when the field is created (one thread per ray), and k_trace_sample starts
// from them. (Until round 3 every thread replayed the additions up to its k0: 250 steps on average, ~45 % of the kernel's
// vector instructions.) segPos[(segment * 3 + component) * R + ray].
__global__ __launch_bounds__(256) void k_trace_segpos(TracerParams tp, int W, int R, float* __restrict__ segPos) {
    const int ray = blockIdx.x * 256 + threadIdx.x;
    if (ray >= R) return;
    const int x = ray % W, y = ray / W;
    Vec3 pos = tp.getStart(x, y);
    const Vec3 step = tp.getInc(x, y);
    for (unsigned int k = 0, seg = 0; k <= tp.steps; ++k) {
        if (k % kTraceSeg == 0) {
            float* q = segPos + (size_t)seg * 3 * R + ray;
            q[0] = pos.x; q[(size_t)R] = pos.y; q[2 * (size_t)R] = pos.z;
            ++seg;
        }
        pos = pos + step;
    }
}

__global__ __launch_bounds__(256 * kTraceSegsPerBlock) void k_trace_sample(const float* __restrict__ ct, int nx, int ny, int nz, LutView lut,
                                                       TracerParams tp, int W, int H, float* __restrict__ bevDensity,
                                                       float* __restrict__ spTerm, float* __restrict__ huBuf,
                                                       float* __restrict__ bevRrl, float rRlScale, FieldState* st,
                                                       const float* __restrict__ segPos) {
    extern __shared__ float sLut[];
    float* sDensity = sLut;
    float* sSp = sLut + lut.nDensity;
    const int tid = threadIdx.y * 256 + threadIdx.x;
    constexpr int nT = 256 * kTraceSegsPerBlock;
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) resetFieldScalars(st);     // (the scan, next launch, accumulates into them)
    const int ray = blockIdx.x * 256 + threadIdx.x;
    const int x = ray % W, y = ray / W;
    const size_t memStep = (size_t)W * H;
    const unsigned int k0 = min((blockIdx.y * kTraceSegsPerBlock + threadIdx.y) * kTraceSeg, tp.steps);
    const unsigned int k1 = min(k0 + kTraceSeg, tp.steps);

    // the LUT rows travel to LDS while the start position of the segment is being accumulated
    constexpr int kLutRegs = 12;                                     // covers 2 x 3072 entries in registers; longer tables loop
    float rl[kLutRegs];
    const int nLut = lut.nDensity + lut.nSp;
#pragma unroll
    for (int j = 0; j < kLutRegs; ++j) {
        const int i = tid + nT * j;
        rl[j] = i < lut.nDensity ? lut.density[i] : (i < nLut ? lut.sp[i - lut.nDensity] : 0.0f);
    }
    // position at the segment's first step (k_trace_segpos: the serial walk's own value; texel-centre +0.5 of kernel_wrapper.cu:142
    // is implicit in the sampler)
    const unsigned int seg = blockIdx.y * kTraceSegsPerBlock + threadIdx.y;
    Vec3 pos = v3(0.0f, 0.0f, 0.0f);
    if (k0 < k1) {
        const float* q = segPos + (size_t)seg * 3 * memStep + ray;
        pos = v3(q[0], q[memStep], q[2 * memStep]);
    }
    const Vec3 step = tp.getInc(x, y);
    const float stepLen = tp.stepLen(x, y);
#pragma unroll
    for (int j = 0; j < kLutRegs; ++j) { const int i = tid + nT * j; if (i < nLut) sLut[i] = rl[j]; }
    for (int i = tid + nT * kLutRegs; i < nLut; i += nT) sLut[i] = i < lut.nDensity ? lut.density[i] : lut.sp[i - lut.nDensity];
    __syncthreads();
    size_t idx = (size_t)k0 * memStep + ray;
    auto emit = [&](float huPlus1000) {
        huBuf[idx] = huPlus1000;
        const float density = sample1dClamp(sDensity, lut.nDensity, huPlus1000 * tp.densityScale);
        bevDensity[idx] = density;
        spTerm[idx] = stepLen * sample1dClamp(sSp, lut.nSp, huPlus1000 * tp.spScale);
        // density * (1/X0 per unit density): the radiation-length factor of the scatter term (kernel_wrapper.cu:285-290) does
        // not depend on the energy layer, so it is evaluated here once per (ray, step) instead of once per layer in k_fill
        bevRrl[idx] = density * sample1dClamp(lut.rrl, lut.nRrl, density * rRlScale);
        idx += memStep;
    };
    // (Measured and dropped: the corner loads of three steps requested together — 74 registers, six waves per SIMD instead of eight:
    //  the stage 78 us against 66. What hides the round trips here is the number of resident waves.)
    unsigned int i = k0;
    for (; i < k1; ++i) {
        emit(sample3dBorder(ct, nx, ny, nz, pos.x, pos.y, pos.z));
        pos = pos + step;
    }
}

// k_trace_sample for beams that run along the CT x axis (gantry near 90 / 270 degrees): there ray-adjacent lanes sample
// CT voxels one whole slice apart and every load of a wave touches 64 lines in 64 slices (measured 0.22 ms against 0.077 ms
// at 0 degrees). Here one wave walks ONE ray with its lanes on consecutive steps (lane l takes steps l, l + 64, ...), so a
// load touches a few consecutive lines; the terms cross an LDS tile [ray][step] and leave with lanes along the rays again
// (16 rays = 64-byte runs, step-major like the plain kernel). Positions are the same serial `pos += step` sequence.
// Oblique beams: a mapping with 8 rays x 8 steps per wave (compact in the rotated plane) measured 0.126 ms at every angle,
// never better than the better of the two lane directions (plain: 0.076 ms at 0 deg, 0.130 at 30, 0.166 at 45; this one: 0.135
// at 30, 0.128 at 45, 0.100 at 90), so the host just picks between those two.
// (The lanes of this kernel walk the whole ray to get from one of their steps to the next; starting from k_trace_segpos's table
//  instead — one more memory round trip in front of the samples — measured no gain here: 0.086 against 0.091 ms at 90 deg, 0.131
//  against 0.120 at 45.)
constexpr int kTrRays = 16, kTrSteps = 512, kTrPitch = kTrSteps + 4;
__global__ __launch_bounds__(64 * kTrRays) void k_trace_sample_t(const float* __restrict__ ct, int nx, int ny, int nz, LutView lut,
                                                                  TracerParams tp, int W, int H, float* __restrict__ bevDensity,
                                                                  float* __restrict__ spTerm, float* __restrict__ huBuf,
                                                                  float* __restrict__ bevRrl, float rRlScale, FieldState* st) {
    extern __shared__ float sLut[];
    float* sDensity = sLut;
    float* sSp = sLut + lut.nDensity;
    const int nLut = lut.nDensity + lut.nSp;
    float* tile = sLut + nLut;                                       // [hu, density, sp][kTrRays][kTrPitch]
    if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) resetFieldScalars(st);
    constexpr int nThreads = 64 * kTrRays;
    const int R = W * H;
    const int lane = threadIdx.x, wv = threadIdx.y, tid = wv * 64 + lane;
    const int ray0 = blockIdx.x * kTrRays;
    // this lane's ray of the block, its first step of a pass, steps per trip, trips per pass
    const int myRay = wv, myStep0 = lane;
    constexpr int kStride = 64, kTrips = kTrSteps / kStride;
    const int ray = min(ray0 + myRay, R - 1);                        // (a last partial block repeats the last ray, stores nothing for it)
    const int x = ray % W, y = ray / W;
    const size_t memStep = (size_t)R;
    for (int i = tid; i < nLut; i += nThreads) sLut[i] = i < lut.nDensity ? lut.density[i] : lut.sp[i - lut.nDensity];
    Vec3 pos = tp.getStart(x, y);
    const Vec3 step = tp.getInc(x, y);
    const float stepLen = tp.stepLen(x, y);
    for (int i = 0; i < myStep0; ++i) pos = pos + step;              // same float sequence as the serial walk
    __syncthreads();
    constexpr int plane = kTrRays * kTrPitch;
    float* tHu = tile + myRay * kTrPitch + myStep0, *tDe = tHu + plane, *tSp = tDe + plane;
    const int oRay = tid % kTrRays, oStep = tid / kTrRays;           // write-out: the rays of one step are neighbours (oStep < 64)
    for (unsigned int base = 0; base < tp.steps; base += kTrSteps) {
        for (int j = 0; j < kTrips; ++j) {
            const unsigned int k = base + myStep0 + kStride * j;
            if (k < tp.steps) {
                const float huPlus1000 = sample3dBorder(ct, nx, ny, nz, pos.x, pos.y, pos.z);
                tHu[kStride * j] = huPlus1000;
                tDe[kStride * j] = sample1dClamp(sDensity, lut.nDensity, huPlus1000 * tp.densityScale);
                tSp[kStride * j] = stepLen * sample1dClamp(sSp, lut.nSp, huPlus1000 * tp.spScale);
            }
            if (k + kStride < tp.steps)                              // on to this lane's next step
                for (int i = 0; i < kStride; ++i) pos = pos + step;
        }
        __syncthreads();
        const float* oHu = tile + oRay * kTrPitch, *oDe = oHu + plane, *oSp = oDe + plane;
        if (ray0 + oRay < R)
            for (int sl = oStep; sl < kTrSteps; sl += 64) {
                const unsigned int k = base + sl;
                if (k < tp.steps) {
                    const size_t idx = (size_t)k * memStep + ray0 + oRay;
                    huBuf[idx] = oHu[sl];
                    const float density = oDe[sl];
                    bevDensity[idx] = density;
                    spTerm[idx] = oSp[sl];
                    bevRrl[idx] = density * sample1dClamp(lut.rrl, lut.nRrl, density * rRlScale);   // (see k_trace_sample)
                }
            }
        __syncthreads();
    }
}

// k_trace_sample for OBLIQUE beams (gantry 35 - 65 degrees about the CT y axis, and their mirror images): there neither lane direction
// of the two kernels above is coherent — 64 rays in x, or 64 steps of one ray, both cross a CT slice per lane or nearly (0.12 - 0.15 ms
// at 45 degrees against 0.057 at 0). But the samples (ray x + j, step k + b j), j = 0, 1, ..., for the right small integer b, lie
// along CT x within ONE slice and row (at 45 degrees, b = 1: exactly): a run of 32 lanes touches a handful of cache lines per corner
// instead of 32. A block is a region of 32 rays (one ray row) x 128 steps; its 4096 samples are taken along those diagonals (ray j,
// step (d + b j) mod 128 for diagonal d), cross an LDS tile [step][ray] and leave step-major with lanes along the rays, like
// k_trace_sample_t. The host picks b (-3 .. 3) that minimises the drift across slices per lane, from the field's geometry, and this
// kernel when that drift is well below both other kernels'. Positions: k_trace_segpos's table + at most kTraceSeg - 1 additions.
constexpr int kTdRays = 32, kTdSteps = 128, kTdPitch = 34, kTdThreads = 1024;   // (pitch 34: the diagonal writes fall into different banks for odd and even b)
__global__ __launch_bounds__(kTdThreads) void k_trace_sample_d(const float* __restrict__ ct, int nx, int ny, int nz, LutView lut,
                                                               TracerParams tp, int W, int H, float* __restrict__ bevDensity,
                                                               float* __restrict__ spTerm, float* __restrict__ huBuf,
                                                               float* __restrict__ bevRrl, float rRlScale, FieldState* st,
                                                               const float* __restrict__ segPos, int diagB) {
    extern __shared__ float sLut[];
    float* sDensity = sLut;
    float* sSp = sLut + lut.nDensity;
    const int nLut = lut.nDensity + lut.nSp;
    float* tile = sLut + nLut;                                       // [hu, density, sp][kTdSteps][kTdPitch]
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) resetFieldScalars(st);
    const int x0 = blockIdx.x * kTdRays, y = blockIdx.y;
    const unsigned int k0 = blockIdx.z * kTdSteps;
    const size_t memStep = (size_t)W * H;
    for (int i = tid; i < nLut; i += kTdThreads) sLut[i] = i < lut.nDensity ? lut.density[i] : lut.sp[i - lut.nDensity];
    __syncthreads();
    constexpr int plane = kTdSteps * kTdPitch;
#pragma unroll
    for (int u = 0; u < kTdRays * kTdSteps / kTdThreads; ++u) {
        const int s = tid + u * kTdThreads;
        const int j = s % kTdRays, d = s / kTdRays;
        const int row = (((d + diagB * j) % kTdSteps) + kTdSteps) % kTdSteps;      // step of the region
        const unsigned int k = k0 + row;
        const int x = x0 + j;
        if (k < tp.steps) {
            const int ray = y * W + x;
            const unsigned int seg = k / kTraceSeg, r = k - seg * kTraceSeg;
            const float* q = segPos + (size_t)seg * 3 * memStep + ray;
            Vec3 pos = v3(q[0], q[memStep], q[2 * memStep]);
            const Vec3 step = tp.getInc(x, y);
            for (unsigned int i = 0; i < r; ++i) pos = pos + step;   // same float sequence as the serial walk
            const float huPlus1000 = sample3dBorder(ct, nx, ny, nz, pos.x, pos.y, pos.z);
            float* t = tile + row * kTdPitch + j;
            t[0] = huPlus1000;
            t[plane] = sample1dClamp(sDensity, lut.nDensity, huPlus1000 * tp.densityScale);
            t[2 * plane] = tp.stepLen(x, y) * sample1dClamp(sSp, lut.nSp, huPlus1000 * tp.spScale);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kTdRays * kTdSteps / kTdThreads; ++u) {
        const int s = tid + u * kTdThreads;
        const int j = s % kTdRays, row = s / kTdRays;
        const unsigned int k = k0 + row;
        if (k < tp.steps) {
            const size_t idx = (size_t)k * memStep + (size_t)y * W + x0 + j;
            const float* t = tile + row * kTdPitch + j;
            huBuf[idx] = t[0];
            const float density = t[plane];
            bevDensity[idx] = density;
            spTerm[idx] = t[2 * plane];
            bevRrl[idx] = density * sample1dClamp(lut.rrl, lut.nRrl, density * rRlScale);   // (see k_trace_sample)
        }
    }
}

// The sums are serial per ray (float order of the reference walk, kernel_wrapper.cu:147-186), so only R/64 serial
// chains of 64 lanes exist: a plain one-wave-per-64-rays walk keeps ~2 MB of loads in flight and is bound by memory
// latency (measured 57 us for 51 MB). Here a block of kScanWaves waves serves 64 rays: ALL waves stream the next chunk of
// kScanChunk steps (hu and stepLen*SP terms) into registers and then LDS, while three of them walk the current chunk
// out of LDS, one serial chain each:
//   wave 0: cumulSp (WEPL; written back into the chunk, then stored by all waves)
//   wave 1: cumulHu -> beforeFirstInside (:173-176)        wave 2: hu > 150 -> lastInside (:177-180)
// (Round 3, measured and dropped: 32 rays per block — 264 blocks instead of 132, the two chains in the halves of one wave. The
//  kernel is bound by the length of the serial chains, not by per-CU bandwidth: 27 -> 60 us.)
constexpr int kScanWaves = 16, kScanChunk = 128, kScanPerWave = kScanChunk / kScanWaves;
__global__ __launch_bounds__(64 * kScanWaves) void k_trace_scan(const float* __restrict__ huBuf, float* __restrict__ bevCumulSp, int W, int H,
                                                                 unsigned int steps, int* __restrict__ firstInside, int* __restrict__ firstOutside,
                                                                 FieldState* st, float* __restrict__ blockWeplMin, ResetJob reset) {
    extern __shared__ float sScan[];                                 // two buffers [hu, sp][kScanChunk][64]: one is walked and stored while the next chunk is staged into the other
    const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int ray = blockIdx.x * 64 + lane;
    const size_t memStep = (size_t)W * H;
    long long* dbg = reset.scanDbg ? reset.scanDbg + 8 * (size_t)blockIdx.x : nullptr;
    int dbgN = 0;
    auto stamp = [&]() { if (dbg && wv == 0 && lane == 0 && dbgN < 8) dbg[dbgN++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    // the waves without a serial chain reset the per-layer records, the tile-radius bytes and the dose rectangles (K0)
    if (wv >= 3) resetFieldArrays(reset, ((size_t)blockIdx.x * (kScanWaves - 3) + (wv - 3)) * 64 + lane, (size_t)gridDim.x * (kScanWaves - 3) * 64);
    auto sHu = [&](int b, int i) -> float& { return sScan[((2 * b) * kScanChunk + i) * 64 + lane]; };
    auto sSp = [&](int b, int i) -> float& { return sScan[((2 * b + 1) * kScanChunk + i) * 64 + lane]; };
    float rHu[kScanPerWave], rSp[kScanPerWave];
    auto fetch = [&](unsigned int c0) {                              // this wave's kScanPerWave steps of the chunk starting at c0
#pragma unroll
        for (int j = 0; j < kScanPerWave; ++j) {
            const unsigned int i = c0 + wv * kScanPerWave + j;
            rHu[j] = i < steps ? huBuf[ray + (size_t)i * memStep] : 0.0f;
            rSp[j] = i < steps ? bevCumulSp[ray + (size_t)i * memStep] : 0.0f;   // holds stepLen*SP(hu) from k_trace_sample
        }
    };
    // Only the two running sums are serial. The two index searches of the reference walk — the last step whose running HU sum is below
    // 150 (:173-176), the last step whose HU is above 150 (:177-180) — are maxima over steps: every wave takes them over ITS steps (the
    // raw HU values as it stages them, the running sums once wave 1 has written them back), and the waves' partial maxima meet at the end.
    int beforeFirstInside = -1, lastInside = -1;                     // this wave's partial maxima (lane = ray)
    auto stage = [&](int buf, unsigned int cBase) {
#pragma unroll
        for (int j = 0; j < kScanPerWave; ++j) {
            sHu(buf, wv * kScanPerWave + j) = rHu[j]; sSp(buf, wv * kScanPerWave + j) = rSp[j];
            lastInside = max(lastInside, rHu[j] > 150.0f ? (int)(cBase + wv * kScanPerWave + j) : -1);   // (zeros beyond the last step never pass)
        }
    };
    float cumulSp = 0.0f, cumulHuPlus1000 = 0.0f;
    fetch(0);
    stage(0, 0u);
    __syncthreads();
    stamp();
    int buf = 0;
    for (unsigned int c0 = 0; c0 < steps; c0 += kScanChunk, buf ^= 1) {
        if (c0 + kScanChunk < steps) fetch(c0 + kScanChunk);         // in flight during the walks below
        // (steps past the end were staged as zeros: they change no sum and set no index)
        constexpr int kU = 16;                                       // LDS reads issued ahead of the serial adds
        if (wv == 0) {
            // (requesting the next batch's LDS reads before this batch's chain of additions measured slower: 7.1 k against 6.0 k cycles per chunk)
            for (int j0 = 0; j0 < kScanChunk; j0 += kU) {
                float v[kU];
#pragma unroll
                for (int j = 0; j < kU; ++j) v[j] = sSp(buf, j0 + j);
#pragma unroll
                for (int j = 0; j < kU; ++j) { cumulSp += v[j]; sSp(buf, j0 + j) = cumulSp; }
            }
        } else if (wv == 1) {
            // (until round 3 this wave also compared every sum with 150 and a third wave searched the raw values: 63 cycles per step,
            //  56 % of the kernel — clock stamps, tools/scan_dbg.py)
            for (int j0 = 0; j0 < kScanChunk; j0 += kU) {
                float v[kU];
#pragma unroll
                for (int j = 0; j < kU; ++j) v[j] = sHu(buf, j0 + j);
#pragma unroll
                for (int j = 0; j < kU; ++j) { cumulHuPlus1000 += v[j]; sHu(buf, j0 + j) = cumulHuPlus1000; }
            }
        }
        ldsBarrier();                                                // chunk walked
        stamp();
        // The next chunk goes from the registers into the OTHER buffer first: this wait is for loads issued a walk ago. (With one
        // buffer the staging came after the stores of the walked chunk, and its wait for the loads — one counter for loads and
        // stores, in order — stood through the stores' whole latency: 5.4 k cycles per 256-step chunk, clock stamps.)
        if (c0 + kScanChunk < steps) stage(buf ^ 1, c0 + kScanChunk);
        float wOut[kScanPerWave], hOut[kScanPerWave];                // (all LDS reads first: one round trip, not one per step)
#pragma unroll
        for (int j = 0; j < kScanPerWave; ++j) { wOut[j] = sSp(buf, wv * kScanPerWave + j); hOut[j] = sHu(buf, wv * kScanPerWave + j); }
#pragma unroll
        for (int j = 0; j < kScanPerWave; ++j) {                     // all waves store the chunk's WEPL
            const unsigned int i = c0 + wv * kScanPerWave + j;
            if (i < steps) {
                if (hOut[j] < 150.0f) beforeFirstInside = max(beforeFirstInside, (int)i);
                bevCumulSp[ray + (size_t)i * memStep] = wOut[j];
            }
        }
        // sliceMinVar<float> (kernel_wrapper.cuh:215-244, launch kernel_wrapper.cu:788), first level: this block's 64 rays; k_plan takes
        // the minimum over the blocks (one value per block and step instead of a pass over all of WEPL). Transposed: lane = step, the 64
        // rays of the step read from the chunk in a rotated order (lane l starts at ray l: 64 banks) — 64 reads and minima per lane and
        // one coalesced store per wave, against a cross-lane reduction and a one-lane store per step.
        if (wv >= 2 && wv < 2 + kScanChunk / 64) {
            const int stepInChunk = (wv - 2) * 64 + lane;
            const float* col = sScan + (size_t)((2 * buf + 1) * kScanChunk + stepInChunk) * 64;
            float m = col[lane];
#pragma unroll 16
            for (int r = 1; r < 64; ++r) { const float t = col[(lane + r) & 63]; m = t < m ? t : m; }
            if (c0 + stepInChunk < steps) blockWeplMin[(size_t)blockIdx.x * steps + c0 + stepInChunk] = m;
        }
        stamp();
        ldsBarrier();                                                // next chunk staged; this one stored: its buffer is free
        stamp();
    }
    // the waves' partial maxima meet: [wave][ray] in the (now free) chunk buffer
    ldsBarrier();
    int* sPart = reinterpret_cast<int*>(sScan);
    sPart[wv * 64 + lane] = beforeFirstInside;
    sPart[(kScanWaves + wv) * 64 + lane] = lastInside;
    ldsBarrier();
    if (wv == 1) {
        int v = -1;
#pragma unroll
        for (int w = 0; w < kScanWaves; ++w) v = max(v, sPart[w * 64 + lane]);
        firstInside[ray] = v + 1;
        const int mn = waveMinI(v + 1);
        if (lane == 0) atomicMin(&st->beamFirstInside, mn);
    } else if (wv == 2) {
        int v = -1;
#pragma unroll
        for (int w = 0; w < kScanWaves; ++w) v = max(v, sPart[(kScanWaves + w) * 64 + lane]);
        firstOutside[ray] = v + 1;
        const int mx = waveMaxI(v + 1);
        if (lane == 0) atomicMax(&st->beamFirstOutside, mx);
    }
}

// ------------------------------------------------------------------------------------------------
// Entry plane of the beam (kernel_wrapper.cu:784, :838-849): depth of the first step inside the patient, the pixel spacing factors
// there and a layer's spot sigma there. Evaluated by k_plan (which records them) AND by the spot -> ray convolution itself, with
// these same expressions — so that the convolution needs nothing k_plan writes and the two can share a launch (k_plan_conv).
struct EntryGeom { float entryZ, pxSpMultX, pxSpMultY; };
__device__ inline EntryGeom entryGeom(int beamFirstInside, const FieldConst& fc) {
    EntryGeom e;
    e.entryZ = ((float)beamFirstInside) * fc.rayRes[2] + fc.rayOffset[2];
    e.pxSpMultX = 1.0f - e.entryZ / fc.sourceDist[0];
    e.pxSpMultY = 1.0f - e.entryZ / fc.sourceDist[1];
    return e;
}
__device__ inline float entrySigma(const LayerPlan& p, float spotSigma, float entryZ, const FieldConst& fc) {
    float s = sqrtf(p.airCoefA * entryZ * entryZ + p.airCoefB * entryZ + spotSigma * spotSigma);
    if (fc.nuclearCorr == 3) s = 0.97f * s;                          // GAUSS_FIT, kernel_wrapper.cu:842-847
    return s;
}

// ------------------------------------------------------------------------------------------------
// K2: device-side plan = the host cut-off logic of kernel_wrapper.cu:784,792-802,829-849,923-924. One workgroup of nT threads
// (tid = its linear thread index): its own launch (k_plan) or one block of k_plan_conv.
__device__ inline void planBody(FieldState* st, LayerPlan* layers, const float* __restrict__ blockWeplMin, int nScanBlocks,
                                int* __restrict__ weplMinBits, const FieldConst& fc, const int tid, const int nT) {
    __shared__ float weplMin[kMaxSteps];
    __shared__ float sPart[8][512];          // partial minima (steps <= 512: up to 8 threads per group of four steps)
    __shared__ int sGuaranteed;
    __shared__ float sEntryZ;
    // sliceMinVar<float>, second level: smallest WEPL of every step over the scan's blocks. blockWeplMin is [block][step]: a thread takes
    // FOUR consecutive steps (one 16-byte load per block) of a share of the blocks, 8 loads in flight (a dependent load here is a
    // full round trip of a single workgroup: this is latency, not bandwidth).
    {
        const float inf = __int_as_float(0x7f800000);
        if ((fc.S & 3) == 0 && fc.S <= 512) {
            const int nQuads = fc.S >> 2;
            const int nParts = max(1, min(8, nT / nQuads));
            for (int idx = tid; idx < nQuads * nParts; idx += nT) {
                const int q = idx % nQuads, part = idx / nQuads;
                const int b0 = (int)((long long)nScanBlocks * part / nParts), b1 = (int)((long long)nScanBlocks * (part + 1) / nParts);
                float4 m = make_float4(inf, inf, inf, inf);
                for (int b = b0; b < b1; b += 8) {
                    float4 t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        t[u] = b + u < b1 ? *reinterpret_cast<const float4*>(blockWeplMin + (size_t)(b + u) * fc.S + 4 * q) : make_float4(inf, inf, inf, inf);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { m.x = t[u].x < m.x ? t[u].x : m.x; m.y = t[u].y < m.y ? t[u].y : m.y; m.z = t[u].z < m.z ? t[u].z : m.z; m.w = t[u].w < m.w ? t[u].w : m.w; }
                }
                sPart[part][4 * q] = m.x; sPart[part][4 * q + 1] = m.y; sPart[part][4 * q + 2] = m.z; sPart[part][4 * q + 3] = m.w;
            }
            __syncthreads();
            for (int s0 = tid; s0 < fc.S; s0 += nT) {
                float m = sPart[0][s0];
                for (int part = 1; part < nParts; ++part) { const float t = sPart[part][s0]; m = t < m ? t : m; }
                weplMin[s0] = m;
                weplMinBits[s0] = __float_as_int(m);                  // kept for rtd_field_fetch("wepl_min")
            }
        } else {
            for (int s0 = tid; s0 < fc.S; s0 += nT) {
                float m = inf;
                for (int b = 0; b < nScanBlocks; b += 16) {
                    float t[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t[u] = b + u < nScanBlocks ? blockWeplMin[(size_t)(b + u) * fc.S + s0] : inf;
#pragma unroll
                    for (int u = 0; u < 16; ++u) m = t[u] < m ? t[u] : m;
                }
                weplMin[s0] = m;
                weplMinBits[s0] = __float_as_int(m);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        int first = st->beamFirstInside;
        const EntryGeom eg = entryGeom(first, fc);
        int firstPastCutoffAll = findFirstLargerOrdered(weplMin, fc.S, fc.bpDepthCutoff * fc.maxPeakDepth);
        int guaranteed = firstPastCutoffAll < st->beamFirstOutside ? firstPastCutoffAll : st->beamFirstOutside;
        st->firstGuaranteedPassive = guaranteed;
        st->entryZ = eg.entryZ;
        st->pxSpMultX = eg.pxSpMultX;
        st->pxSpMultY = eg.pxSpMultY;
        st->empty = guaranteed > first ? 0 : 1;
        sGuaranteed = guaranteed; sEntryZ = eg.entryZ;
    }
    __syncthreads();
    const float entryZ = sEntryZ;
    for (int l = tid; l < fc.L; l += nT) {
        LayerPlan& p = layers[l];
        p.entrySigmaX = entrySigma(p, p.spotSigmaX, entryZ, fc);
        p.entrySigmaY = entrySigma(p, p.spotSigmaY, entryZ, fc);
        unsigned int localAfterLast = (unsigned int)findFirstLargerOrdered(weplMin, fc.S, fc.bpDepthCutoff * p.peakDepth);
        unsigned int g = (unsigned int)sGuaranteed;
        p.afterLast = (int)(localAfterLast < g ? localAfterLast : g);
    }
    __syncthreads();
    // k_fill's walks — (layer, role): role 0 the sigma walk, role 1 the dose walk — ranked by descending cost: steps of the layer x
    // a measured per-step weight of the role (155 : 100). Stable rank by counting; 2 L <= 512 entries.
    for (int p = tid; p < 2 * fc.L; p += nT) {
        const int a = layers[p >> 1].afterLast * ((p & 1) ? 100 : 155);
        int rank = 0;
        for (int u = 0; u < 2 * fc.L; ++u) { const int au = layers[u >> 1].afterLast * ((u & 1) ? 100 : 155); rank += (au > a || (au == a && u < p)) ? 1 : 0; }
        st->fillItems[rank] = (unsigned short)p;
    }
}
__global__ __launch_bounds__(1024) void k_plan(FieldState* st, LayerPlan* layers, const float* __restrict__ blockWeplMin, int nScanBlocks,
                                               int* __restrict__ weplMinBits, FieldConst fc) {
    planBody(st, layers, blockWeplMin, nScanBlocks, weplMinBits, fc, (int)threadIdx.x, (int)blockDim.x);
}

#define RTD_DM_FN __device__ inline
#include "../../include/rtd_detmath.h"

// ------------------------------------------------------------------------------------------------
// K3/K4: spot -> ray weights, separable erf-integrated Gaussian resampling (gpu_convolution_2d.cu:16-59). The error
// function is rtd_erf_det (rtd_detmath.h): bit-reproducible, so the RAY_WEIGHT_CUTOFF liveness test downstream is too.
__global__ void k_conv_x(const float* __restrict__ in, float* __restrict__ out, const LayerPlan* __restrict__ layers,
                         const FieldState* __restrict__ st, FieldConst fc) {
    const int idxY = blockDim.y * blockIdx.y + threadIdx.y;
    const int z = blockIdx.z;
    const int inWidth = fc.spotNx, height = fc.spotNy, outWidth = fc.W;
    const float inOutDelta = fc.spotDelta[0] / fc.rayRes[0];
    const float inOutOffset = (fc.spotOffset[0] - fc.rayOffset[0]) / fc.rayRes[0];
    const EntryGeom eg = entryGeom(st->beamFirstInside, fc);
    const float pixelSp = fc.rayRes[0] * eg.pxSpMultX;
    const float cut = fc.convSigmaCutoff;
    if (idxY < height) {
        const int outIdxX = blockDim.x * blockIdx.x + threadIdx.x;
        float res = 0.0f;
        float sigmaEff = entrySigma(layers[z], layers[z].spotSigmaX, eg.entryZ, fc) / pixelSp;
        float rSigmaEff = (1.0f / sqrtf(2.0f)) / sigmaEff;
        int cur = f2iSat(ceilf(((float)outIdxX - (cut * sigmaEff + 0.5f) - inOutOffset) / inOutDelta));
        cur = cur < 0 ? 0 : cur;   // spots left of the map contribute nothing: skip them (bounded loop, same result)
        float dist = (float)cur * inOutDelta + inOutOffset - (float)outIdxX;
        while (dist < (cut * sigmaEff + 0.5f) && cur < inWidth) {
            if (cur >= 0 && cur < inWidth)
                res += 0.5f * (rtd_erf_det((dist + 0.5f) * rSigmaEff) - rtd_erf_det((dist - 0.5f) * rSigmaEff))
                       * in[(size_t)z * inWidth * height + (size_t)idxY * inWidth + cur];
            ++cur;
            dist = (float)cur * inOutDelta + inOutOffset - (float)outIdxX;
        }
        out[(size_t)z * outWidth * height + (size_t)idxY * outWidth + outIdxX] = res;
    }
}
__global__ void k_conv_y(const float* __restrict__ in, float* __restrict__ out, const LayerPlan* __restrict__ layers,
                         const FieldState* __restrict__ st, FieldConst fc) {
    const int idxX = blockDim.x * blockIdx.x + threadIdx.x;
    const int z = blockIdx.z;
    const int width = fc.W, inHeight = fc.spotNy, outHeight = fc.H;
    const float inOutDelta = fc.spotDelta[1] / fc.rayRes[1];
    const float inOutOffset = (fc.spotOffset[1] - fc.rayOffset[1]) / fc.rayRes[1];
    const EntryGeom eg = entryGeom(st->beamFirstInside, fc);
    const float pixelSp = fc.rayRes[1] * eg.pxSpMultY;
    const float cut = fc.convSigmaCutoff;
    if (idxX < width) {
        const int outIdxY = blockDim.y * blockIdx.y + threadIdx.y;
        float res = 0.0f;
        float sigmaEff = entrySigma(layers[z], layers[z].spotSigmaY, eg.entryZ, fc) / pixelSp;
        float rSigmaEff = (1.0f / sqrtf(2.0f)) / sigmaEff;
        int cur = f2iSat(ceilf(((float)outIdxY - (cut * sigmaEff + 0.5f) - inOutOffset) / inOutDelta));
        cur = cur < 0 ? 0 : cur;
        float dist = (float)cur * inOutDelta + inOutOffset - (float)outIdxY;
        while (dist < (cut * sigmaEff + 0.5f) && cur < inHeight) {
            if (cur >= 0 && cur < inHeight)
                res += 0.5f * (rtd_erf_det((dist + 0.5f) * rSigmaEff) - rtd_erf_det((dist - 0.5f) * rSigmaEff))
                       * in[(size_t)z * width * inHeight + (size_t)cur * width + idxX];
            ++cur;
            dist = (float)cur * inOutDelta + inOutOffset - (float)outIdxY;
        }
        out[(size_t)z * width * outHeight + (size_t)outIdxY * width + idxX] = res;
    }
}

// K3+K4 in one launch, the x pass staged through LDS: a block owns a 32 x 8 tile of rays of one layer, evaluates the x pass for the
// spot rows its y pass can reach (the same expression per value as k_conv_x: the ray weights stay bit-identical) into an LDS tile
// and runs the y pass from there. The rows of a tile are evaluated again by the tiles above and below (~7x on the bench plan) —
// cheaper than a second launch with its round trip through memory: 10.7 us for the pair of kernels above, 5 us for this one.
// Used whenever the spot map has at most kConvMaxRows rows (LDS tile of 32 floats per row).
constexpr int kConvMaxRows = 384;
// One 32 x 8 tile of rays of layer z: tile (bx, by), thread (tx, ty) of its 256 threads, sInterm = the tile's LDS rows. Contains ONE
// __syncthreads(): every thread of the block calls it (a tile beyond the grid passes by >= gridY and only keeps the barrier).
__device__ inline void convTile(const float* __restrict__ in, float* __restrict__ out, const LayerPlan* __restrict__ layers,
                                const FieldState* __restrict__ st, const FieldConst& fc, const int bx, const int by, const int z,
                                const int tx, const int ty, float* __restrict__ sInterm) {
    const int inWidth = fc.spotNx, inHeight = fc.spotNy, width = fc.W, outHeight = fc.H;
    const float cut = fc.convSigmaCutoff;
    const bool tileIn = by * 8 < outHeight && z < fc.L;
    const LayerPlan& lp = layers[tileIn ? z : 0];
    // the entry plane from the tracer's result itself (same expressions as k_plan: the convolution does not wait for it)
    const EntryGeom eg = entryGeom(st->beamFirstInside, fc);
    // y pass geometry (k_conv_y)
    const float inOutDeltaY = fc.spotDelta[1] / fc.rayRes[1];
    const float inOutOffsetY = (fc.spotOffset[1] - fc.rayOffset[1]) / fc.rayRes[1];
    const float pixelSpY = fc.rayRes[1] * eg.pxSpMultY;
    const float sigmaEffY = entrySigma(lp, lp.spotSigmaY, eg.entryZ, fc) / pixelSpY;
    const float rSigmaEffY = (1.0f / sqrtf(2.0f)) / sigmaEffY;
    auto firstRow = [&](int outIdxY) {
        int cur = f2iSat(ceilf(((float)outIdxY - (cut * sigmaEffY + 0.5f) - inOutOffsetY) / inOutDeltaY));
        return cur < 0 ? 0 : cur;
    };
    // spot rows the tile's y pass can read: from the first row of its first output row to the end of the last one's loop (both are
    // monotone in the output row for a positive row spacing; otherwise all rows are staged)
    const int oy0 = 8 * by, oy1 = min(oy0 + 8 - 1, outHeight - 1);
    int rowLo = 0, rowHi = inHeight;                                 // rowHi exclusive
    if (inOutDeltaY > 0.0f) {
        rowLo = firstRow(oy0);
        int c = firstRow(oy1);
        float dist = (float)c * inOutDeltaY + inOutOffsetY - (float)oy1;
        while (dist < (cut * sigmaEffY + 0.5f) && c < inHeight) { ++c; dist = (float)c * inOutDeltaY + inOutOffsetY - (float)oy1; }
        rowHi = min(c, inHeight);
        rowLo = min(rowLo, rowHi);
    }
    const int nRows = tileIn ? rowHi - rowLo : 0;
    // ---- x pass (k_conv_x) for rows [rowLo, rowHi) x the tile's 32 columns ----
    {
        const float inOutDelta = fc.spotDelta[0] / fc.rayRes[0];
        const float inOutOffset = (fc.spotOffset[0] - fc.rayOffset[0]) / fc.rayRes[0];
        const float pixelSp = fc.rayRes[0] * eg.pxSpMultX;
        const int tid = ty * 32 + tx;
        for (int v = tid; v < nRows * 32; v += 256) {
            const int idxY = rowLo + (v >> 5);
            const int outIdxX = 32 * bx + (v & 31);
            float res = 0.0f;
            float sigmaEff = entrySigma(lp, lp.spotSigmaX, eg.entryZ, fc) / pixelSp;
            float rSigmaEff = (1.0f / sqrtf(2.0f)) / sigmaEff;
            int cur = f2iSat(ceilf(((float)outIdxX - (cut * sigmaEff + 0.5f) - inOutOffset) / inOutDelta));
            cur = cur < 0 ? 0 : cur;
            float dist = (float)cur * inOutDelta + inOutOffset - (float)outIdxX;
            while (dist < (cut * sigmaEff + 0.5f) && cur < inWidth) {
                if (cur >= 0 && cur < inWidth)
                    res += 0.5f * (rtd_erf_det((dist + 0.5f) * rSigmaEff) - rtd_erf_det((dist - 0.5f) * rSigmaEff))
                           * in[(size_t)z * inWidth * inHeight + (size_t)idxY * inWidth + cur];
                ++cur;
                dist = (float)cur * inOutDelta + inOutOffset - (float)outIdxX;
            }
            sInterm[v] = res;
        }
    }
    __syncthreads();
    // ---- y pass (k_conv_y) from the LDS tile ----
    const int idxX = 32 * bx + tx;
    const int outIdxY = oy0 + ty;
    if (tileIn && idxX < width && outIdxY < outHeight) {
        float res = 0.0f;
        int cur = firstRow(outIdxY);
        float dist = (float)cur * inOutDeltaY + inOutOffsetY - (float)outIdxY;
        while (dist < (cut * sigmaEffY + 0.5f) && cur < inHeight) {
            if (cur >= 0 && cur < inHeight)
                res += 0.5f * (rtd_erf_det((dist + 0.5f) * rSigmaEffY) - rtd_erf_det((dist - 0.5f) * rSigmaEffY))
                       * sInterm[(cur - rowLo) * 32 + tx];
            ++cur;
            dist = (float)cur * inOutDeltaY + inOutOffsetY - (float)outIdxY;
        }
        out[(size_t)z * width * outHeight + (size_t)outIdxY * width + idxX] = res;
    }
}
__global__ __launch_bounds__(256) void k_conv(const float* __restrict__ in, float* __restrict__ out, const LayerPlan* __restrict__ layers,
                                               const FieldState* __restrict__ st, FieldConst fc) {
    extern __shared__ float sInterm[];                               // [row - rowLo][32]
    convTile(in, out, layers, st, fc, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)threadIdx.x, (int)threadIdx.y, sInterm);
}

// K2 and K3+K4 in ONE launch: neither needs the other (the convolution evaluates the entry plane itself), and each alone is a
// latency-bound launch of a few microseconds on the field's critical path. Blocks of 1024 threads; z < L: four 32 x 8 ray tiles of
// layer z (rows 4 by .. 4 by + 3 of the tile grid), each with its own LDS rows; block (0, 0, L): the plan.
constexpr int kPlanConvMaxRows = 64;                                 // spot rows up to which the four tiles' LDS stays small (4 x 8 KiB)
__global__ __launch_bounds__(1024) void k_plan_conv(const float* __restrict__ in, float* __restrict__ out, LayerPlan* layers, FieldState* st,
                                                    const float* __restrict__ blockWeplMin, int nScanBlocks, int* __restrict__ weplMinBits, FieldConst fc) {
    extern __shared__ float sInterm[];                               // [4 tiles][spotNy][32]
    const int tid = threadIdx.x;
    if ((int)blockIdx.z == fc.L) {
        if (blockIdx.x == 0 && blockIdx.y == 0) planBody(st, layers, blockWeplMin, nScanBlocks, weplMinBits, fc, tid, (int)blockDim.x);
        return;
    }
    const int v = tid >> 8, t = tid & 255;
    convTile(in, out, layers, st, fc, (int)blockIdx.x, 4 * (int)blockIdx.y + v, (int)blockIdx.z, t & 31, t >> 5, sInterm + (size_t)v * fc.spotNy * 32);
}

// ------------------------------------------------------------------------------------------------
// K5: IDD + sigma fill = fillIddAndSigma without NUCLEAR_CORR (kernel_wrapper.cu:190-379), all energy layers in
// one launch, fused with the reductions and the classification that follow it in the reference: layerFirstPassive
// (sliceMaxVar, :952-957) and the per-tile radius class + histogram (tileRadCalc, kernel_wrapper.cuh:256-313).
//
// Block = one (layer, 32x8 classification tile, ROLE) = 4 waves: every ray is walked by TWO threads (of different blocks).
// The reference's step computes two things that share nothing but the liveness of the ray (a function of WEPL alone):
//   role 0, the sigma walk:  residual-range power, betaP, thetaSq and the serial sums sigmaSq / incScat / incincScat /
//                            incDiv (:276-303) -> 1/sigma; per batch of steps the tile's radius class and histogram;
//   role 1, the dose walk:   cumulative-IDD lookup (:269-274), mass and the dose value (:305-322); per batch the rectangle
//                            of the tile's rays that carry dose.
// The serial chains exist once per (ray, layer) — L*R/64 = 2640 waves on C3, 2.6 per SIMD — which is why the single-role form
// of this kernel sat at 61 % of its own instruction-issue bound; two roles double the waves per SIMD and halve each chain.
//
// INDEX WORK IS BIT-EXACT. 1/sigma feeds an integer: the tile minimum is thresholded into the radius class. Everything the
// class depends on is therefore computed with correctly rounded IEEE operations in the reference's order: the power with
// rtd_pow_det (include/rtd_detmath.h, shared with the host-side checker of the test suite — the reference's __powf is a hardware approximation),
// both divisions of betaP / thetaSq with IEEE division, the sums as written. The per-ray 1/sigma that the superposition reads
// for its weights uses the hardware sqrt / reciprocal (<= 2 ulp; the reference builds with -use_fast_math), but the tile's class
// does not come from those values: x -> step/(sqrt2*(sqrt(x)+delta)) is monotone under correct rounding, so the tile
// minimum of the exact 1/sigma equals that function of the tile MAXIMUM of sigmaSq, evaluated once per (step, tile) with
// IEEE sqrt and division.

constexpr int kFillBatch = 8;   // steps per batch: inputs fetched one batch ahead, one block barrier per batch
constexpr int kFillSnakeRounds = 12;   // up to this many walks per CU the walks are dealt in rounds of alternating direction (k_fill's block placement)

// NUCLEAR_CORR arguments of the fill (kernel_wrapper.cu:190-198). The reference constructs its fill parameters with a nuclear
// memory step of 0 (:925), so every step of a ray overwrites the same voxel of the nuclear arrays (:367-373) and what remains
// after a layer's launch is the value of the LAST step, in plane 0. The engine keeps exactly that: one plane per layer, written
// once after the walk.
struct NucFill {
    const int* spotIdx;            // [H][W] the ray's spot on the nuclear grid, -1 if none (kernel_wrapper.cu:878-892)
    const float* rayWeights;       // [L][nucH][nucW] padded spot weights (extendAndPadd, :51-66)
    float* idd; float* rs;         // [L][nucH][nucW] plane 0 of the reference's nuclear arrays after layer l
};

// NUC: NUCLEAR_CORR compiled in (its table lookups and IEEE divisions cost registers: the default build keeps 6 blocks per CU)
template <bool LDS_LUT, bool NUC>
__global__ __launch_bounds__(256) void k_fill(const float* __restrict__ bevDensity, const float* __restrict__ bevCumulSp,
                                               const float* __restrict__ bevRrl,
                                               float* __restrict__ bevIdd, float* __restrict__ bevRSigmaEff,
                                               const float* __restrict__ rayWeights, const int* __restrict__ firstInside,
                                               const int* __restrict__ firstOutside, int* __restrict__ firstPassive,
                                               unsigned char* __restrict__ tileRad, LayerPlan* layers, FieldState* st,
                                               LutView lut, FillGeom fg, FieldConst fc, const float* __restrict__ stepTab,
                                               int* __restrict__ active, int nCU, long long* __restrict__ dbg, NucFill nuc,
                                               unsigned int* __restrict__ sigMin, unsigned int* __restrict__ sigMax, int trackUniform) {
    extern __shared__ float sLutF[];                                 // dose walk: the layer's two cumulative-IDD rows
    // diagnostic build only (RTD_FILL_DEBUG): per walk start / end clock, hardware id, item — no output value depends on it
    // sigma walk: [buffer][step][ray] sigmaSq of the rays with a finite 1/sigma (-1: none) — in the same dynamic LDS as the dose walk's
    // LUT rows (a block is one or the other: 16 KB instead of 16 + 8, a seventh block per CU where the walks outnumber the slots)
    float (*sSig)[kFillBatch][256] = reinterpret_cast<float (*)[kFillBatch][256]>(sLutF);
    __shared__ unsigned long long sDoseMask[2][kFillBatch][4];       // dose walk: [buffer][step][wave] ballot of the rays that carry dose
    __shared__ int sHist[kMaxSuperpR + 2];
    __shared__ int sClassLo[kMaxSuperpR + 2], sClassHi[kMaxSuperpR + 2];   // sigma walk: first / last step of this walk with a tile of that radius class
    __shared__ int sUni;                                             // sigma walk: every tile of this block so far had ONE sigma^2 over its live rays

    // Block placement. The walks differ in cost — the number of steps per layer (150..210 on C3), and a sigma walk is ~1.5 dose
    // walks — while only 2*L*tiles blocks exist (5.2 per CU on C3), so a plain grid leaves the kernel waiting for the CUs that
    // happened to receive the long walks (measured with per-block clock stamps: a CU with five sigma walks of the longest layers
    // took 430 k cycles, one with five dose walks 200 k). When all blocks are co-resident the dispatcher places block b on CU
    // b % nCU (measured: blocks b and b + nCU always share a CU), so the walks, taken in descending order of cost (k_plan), are
    // dealt in rounds of nCU that alternate direction — a CU that got an expensive walk in one round gets a cheap one in the
    // next — with the direction chosen so that the last, partial round (the cheapest walks) lands on the CUs that received the
    // cheapest walks of the last full round. (Performance only: any placement gives the same result.)
    // (More walks than can be co-resident — the reference's water cube: 2560 on 256 CUs — are still dealt this way, up to
    //  kFillSnakeRounds per CU: the first rounds land as described, the rest wherever a slot frees. Measured and dropped there: as many
    //  blocks as fit the GPU at once, each taking walk after walk from a ticket counter in descending order of cost — the loop costs
    //  the kernel 23 registers, 5 blocks per CU instead of 7: 0.325 ms against 0.27.)
    const int nTiles = fc.tilesX * fc.tilesY, nB = 2 * nTiles * fc.L;
    const int tid = threadIdx.y * 32 + threadIdx.x;                  // ray of the tile
    int item = blockIdx.x;
    if (nB <= kFillSnakeRounds * nCU) {
        const int rr = blockIdx.x / nCU, c = blockIdx.x % nCU, nFull = nB / nCU;
        const bool reversed = rr < nFull && ((nFull - 1 - rr) & 1) == 0;       // the last full round: CU 0 gets its cheapest walk
        item = rr * nCU + (reversed ? nCU - 1 - c : c);
    }
    const long long dbgT0 = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const int pr = st->fillItems[item / nTiles];
    const int layer = pr >> 1;
    const int role = pr & 1;                                         // block-uniform: 0 sigma walk, 1 dose walk
    // (the tile is rotated with the walk's index: when the tile count divides the CU count — 64 tiles on 256 CUs, the reference's water
    //  cube — block b and b + nCU would otherwise hold the same tile, and the CUs of the cheap edge tiles would get cheap walks in every round:
    //  measured 1.2 M against 2.6 M block-cycles per CU)
    const int tileNo = (item % nTiles + (item / nTiles) * 5) % nTiles, tileX = tileNo % fc.tilesX, tileY = tileNo / fc.tilesX;
    const int wave = tid >> 6;
    const int x = tileX * kSuperpTileX + threadIdx.x;
    const int y = tileY * kSuperpTileY + threadIdx.y;
    const int W = fc.W, H = fc.H;
    const size_t memStep = (size_t)W * H;
    const size_t layerOff = (size_t)layer * memStep * fc.S;
    const size_t rayIdx = (size_t)y * W + x;
    const unsigned int rayOff = (unsigned int)rayIdx;

    const LayerPlan lp = layers[layer];
    const unsigned int pFirst = (unsigned int)st->beamFirstInside;
    const unsigned int pAfterLast = st->empty ? pFirst : (unsigned int)lp.afterLast;

    // liveness of the ray (:236-243, :308-311): a function of WEPL, the ray weight and the cut-off steps — both roles track it
    bool beamLive = true;
    const int firstIn = firstInside[rayIdx];
    const int fo = firstOutside[rayIdx];
    unsigned int afterLast = (unsigned int)(fo < (int)pAfterLast ? fo : (int)pAfterLast);
    const float rayWeight = rayWeights[(size_t)layer * memStep + rayIdx];
    if (rayWeight < fc.rayWeightCutoff || afterLast < pFirst) { beamLive = false; afterLast = 0; }
    const float cutDepth = lp.peakDepth * fc.bpDepthCutoff;
    float cumulSpOld = 0.0f;
    const float sqrt2 = 1.41421356f;

    if (role == 0) {
        // ================================ sigma walk ================================
        if (tid < kMaxSuperpR + 2) { sHist[tid] = 0; sClassLo[tid] = 0x7fffffff; sClassHi[tid] = -1; }
        if (tid == 0) sUni = trackUniform;                           // 0: the field is known not to be uniform (or not eligible): nothing is tracked
        const float pInv = 0.5649718f, eCoef = 8.639415f;
        // E_s^2 and the empirical widening per NUCLEAR_CORR variant (kernel_wrapper.cu:228-245)
        const float eRefSq = !NUC ? 198.81f : fc.nuclearCorr == 1 ? 190.44f : fc.nuclearCorr == 2 ? 216.09f : fc.nuclearCorr == 3 ? 169.00f : 198.81f;
        const float sigmaDeltaV = !NUC ? 0.21f : fc.nuclearCorr == 1 ? 0.0f : fc.nuclearCorr == 2 ? 0.08f : fc.nuclearCorr == 3 ? 0.06f : 0.21f;
        const int nucIdx = NUC && fc.nuclearCorr ? nuc.spotIdx[rayIdx] : -1;
        const float entrySigmaSq = lp.entrySigmaX * lp.entrySigmaX;  // FillIddAndSigmaParams::getEntrySigmaSq (:925, 4th argument)
        float nucRSigmaEff = __int_as_float(0x7f800000);
        float rSigmaEff = 0.0f, incScat = 0.0f, incincScat = 0.0f;
        float incDiv = lp.sigmaSqAirLin + (2.0f * (float)pFirst - 1.0f) * lp.sigmaSqAirQuad;
        float sigmaSq = -incDiv;
        // Inputs are fetched one batch ahead of the walk (a round trip per batch was 290 cycles per step), in a rolling fashion:
        // as soon as a step has consumed its register slot, the slot takes the load of the step one batch later.
        float spB[kFillBatch], denB[kFillBatch], rrlB[kFillBatch];
        auto fetch1 = [&](int j, unsigned int stepNo) {              // wave-uniform slice base + the lane's ray offset
            spB[j] = 0.0f; denB[j] = 0.0f; rrlB[j] = 0.0f;
            if (stepNo < pAfterLast) {
                spB[j] = (bevCumulSp + (size_t)stepNo * memStep)[rayOff];
                denB[j] = (bevDensity + (size_t)stepNo * memStep)[rayOff];
                rrlB[j] = (bevRrl + (size_t)stepNo * memStep)[rayOff];
            }
        };
#pragma unroll
        for (int j = 0; j < kFillBatch; ++j) fetch1(j, pFirst + j);
        __syncthreads();
        int buf = 0;
        for (unsigned int step0 = pFirst; step0 < pAfterLast; step0 += kFillBatch, buf ^= 1) {
#pragma unroll
            for (int j = 0; j < kFillBatch; ++j) {
                const unsigned int stepNo = step0 + j;
                if (stepNo >= pAfterLast) { sSig[buf][j][tid] = -1.0f; continue; }   // block-uniform
                const float cumulSp = spB[j], density = denB[j], rRl = rrlB[j];
                fetch1(j, stepNo + kFillBatch);
                if (beamLive) {
                    if (cumulSp < lp.peakDepth) {
                        const float resE = eCoef * rtd_pow_det(lp.peakDepth - 0.5f * (cumulSp + cumulSpOld), pInv);
                        const float betaP = resE + 938.3f - 938.3f * 938.3f / (resE + 938.3f);
                        const float thetaSq = eRefSq / (betaP * betaP) * fg.stepLength * rRl;
                        sigmaSq += incScat + incDiv;
                        incincScat += 2.0f * thetaSq * fg.stepLength * fg.stepLength;
                        incScat += incincScat;
                        incDiv += 2.0f * lp.sigmaSqAirQuad;
                    } else {
                        if (!NUC || fc.nuclearCorr != 3) sigmaSq -= 1.5f * (incScat + incDiv) * density;   // (not for GAUSS_FIT, :300-302)
                    }
                    // stepTab[2k] = 0.5*(voxelWidth(k).x + voxelWidth(k).y): per-step constant evaluated once on the host with the
                    // reference's expressions (fill_idd_and_sigma_params.cu:42-46). Hardware sqrt / reciprocal: this value only
                    // weights the superposition; the radius class comes from sigmaSq itself (below).
                    rSigmaEff = stepTab[2 * stepNo] * __builtin_amdgcn_rcpf(sqrt2 * (__builtin_amdgcn_sqrtf(sigmaSq) + sigmaDeltaV));
                    if (NUC && nucIdx >= 0) {                        // :332-341 (IEEE: its tile minimum becomes a radius class too)
                        const float nucSqSigma = sample2dClamp(lut.nucSqSigma, lut.nSamples, lut.nEnergies,
                                                               0.5f * (cumulSp + cumulSpOld) * lp.energyScaleFact, lp.energyIdx);
                        const Vec2 vw = fg.voxelWidth(stepNo);
                        nucRSigmaEff = 0.5f * fc.spotDist * (vw.x + vw.y) / (sqrt2 * sqrtf(sigmaSq + nucSqSigma + entrySigmaSq));
                    }
                    if (cumulSp > cutDepth || stepNo == afterLast) { beamLive = false; afterLast = stepNo; }
                    cumulSpOld = cumulSp;
                }
                float sig = sigmaSq;
                if (!beamLive || (int)stepNo < (firstIn - 1)) { rSigmaEff = __int_as_float(0x7f800000); sig = -1.0f; nucRSigmaEff = __int_as_float(0x7f800000); }
                (bevRSigmaEff + layerOff + (size_t)stepNo * memStep)[rayOff] = rSigmaEff;
                sSig[buf][j][tid] = sig;
            }
            ldsBarrier();                                            // the only barrier of a batch (sSig is double-buffered)
            {   // fused tileRadCalc: radius class of every (layer, step, tile) of the batch, 32 lanes per step
                const int j = tid >> 5, l = tid & 31;                // step of the batch, lane of its 32-lane group
                // (uniform-sigma detection, while the block has seen nothing else: the smallest sigma^2 of the live rays as well —
                //  a block of a heterogeneous field drops this after its first batch)
                const bool uni = sUni != 0;                          // block-uniform (written before the previous batch's barrier)
                const float inf = __int_as_float(0x7f800000);
                float m = sSig[buf][j][l];
                float mn = m >= 0.0f ? m : inf;
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    const float t = sSig[buf][j][l + 32 * k];
                    m = t > m ? t : m;
                    if (uni) { const float tl = t >= 0.0f ? t : inf; mn = tl < mn ? tl : mn; }
                }
                m = -halfWaveMin(-m);                                // lanes 31 / 63 hold the maximum of their 32-lane half
                if (uni) mn = halfWaveMin(mn);
                if (uni && l == 31 && step0 + j < pAfterLast && m >= 0.0f) {
                    if (mn != m) { sUni = 0; st->nonUniform = 1; }
                    else {
                        const size_t si = (size_t)layer * fc.S + step0 + j;
                        atomicMin(&sigMin[si], __float_as_uint(m));  // (sigma^2 >= 0: the bit patterns order like the values)
                        atomicMax(&sigMax[si], __float_as_uint(m));
                    }
                }
                if (l == 31 && step0 + j < pAfterLast) {
                    // tile minimum of 1/sigma (= the reference's minVal, kernel_wrapper.cuh:282-297) from the tile maximum of
                    // sigmaSq with IEEE sqrt and division, then the class exactly as the reference computes it (:300-305)
                    const float minRs = m >= 0.0f ? stepTab[2 * (step0 + j)] / (sqrt2 * (sqrtf(m) + sigmaDeltaV)) : __int_as_float(0x7f800000);
                    int rad = f2iSat(fc.ksSigmaCutoff / (sqrtf(2.0f) * minRs) + 0.5f);
                    rad = rad > kMaxSuperpR + 1 ? kMaxSuperpR + 1 : rad;
                    rad = rad < 0 ? 0 : rad;
                    tileRad[((size_t)layer * fc.S + step0 + j) * nTiles + tileNo] = (unsigned char)rad;
                    atomicAdd(&sHist[rad], 1);
                    atomicMin(&sClassLo[rad], (int)(step0 + j));
                    atomicMax(&sClassHi[rad], (int)(step0 + j));
                }
            }
        }
        firstPassive[(size_t)layer * memStep + rayIdx] = (int)afterLast;
        if (NUC && nucIdx >= 0 && pFirst < pAfterLast) nuc.rs[(size_t)layer * fc.nucW * fc.nucH + nucIdx] = nucRSigmaEff;   // value of the last step (:367-373)
        int mx = waveMaxI((int)afterLast);
        if ((tid & (kWave - 1)) == 0) atomicMax(&layers[layer].layerFirstPassive, mx);
        __syncthreads();
        if (tid < kMaxSuperpR + 2 && sHist[tid] > 0) {
            atomicAdd(&layers[layer].hist[tid], sHist[tid]);
            atomicMin(&layers[layer].classLo[tid], sClassLo[tid]);
            atomicMax(&layers[layer].classHi[tid], sClassHi[tid]);
        }
    } else {
        // ================================ dose walk ================================
        // cumulative IDD: rows floor(energyIdx), floor(energyIdx)+1 (CLAMP) and the row weight are layer constants
        int ey0, ey1; float eay;
        {
            float py = lp.energyIdx, fy = floorf(py);
            eay = py - fy; ey0 = (int)fy; ey1 = ey0 + 1;
            if (!(py >= 0.0f)) { ey0 = 0; ey1 = 0; eay = 0.0f; }
            ey0 = ey0 > lut.nEnergies - 1 ? lut.nEnergies - 1 : ey0;
            ey1 = ey1 > lut.nEnergies - 1 ? lut.nEnergies - 1 : ey1;
        }
        const float* gRow0 = lut.cidd + (size_t)ey0 * lut.nSamples;
        const float* gRow1 = lut.cidd + (size_t)ey1 * lut.nSamples;
        float* sRow0 = sLutF;
        float* sRow1 = sLutF + lut.nSamples;
        if (LDS_LUT) for (int i = tid; i < lut.nSamples; i += 256) { sRow0[i] = gRow0[i]; sRow1[i] = gRow1[i]; }
        float res = 0.0f, cumulDoseOld = 0.0f;
        const int nucIdx = NUC && fc.nuclearCorr ? nuc.spotIdx[rayIdx] : -1;
        const float nucRayWeight = nucIdx >= 0 ? nuc.rayWeights[(size_t)layer * fc.nucW * fc.nucH + nucIdx] : 0.0f;
        float nucRes = 0.0f;
        int actUni = 0x7fffffff;
        float spB[kFillBatch], denB[kFillBatch];
        auto fetch1 = [&](int j, unsigned int stepNo) {
            spB[j] = 0.0f; denB[j] = 0.0f;
            if (stepNo < pAfterLast) {
                spB[j] = (bevCumulSp + (size_t)stepNo * memStep)[rayOff];
                if (!fc.doseToWater) denB[j] = (bevDensity + (size_t)stepNo * memStep)[rayOff];
            }
        };
#pragma unroll
        for (int j = 0; j < kFillBatch; ++j) fetch1(j, pFirst + j);
        __syncthreads();
        int buf = 0;
        for (unsigned int step0 = pFirst; step0 < pAfterLast; step0 += kFillBatch, buf ^= 1) {
            unsigned long long doseMask[kFillBatch];
#pragma unroll
            for (int j = 0; j < kFillBatch; ++j) {
                const unsigned int stepNo = step0 + j;
                doseMask[j] = 0ull;
                if (stepNo >= pAfterLast) continue;                  // block-uniform
                const float cumulSp = spB[j], density = denB[j];
                fetch1(j, stepNo + kFillBatch);
                if (beamLive) {
                    float cumulDose;
                    {   // tex2D(cumulIddTex, ...) :269-274, rows and row weight hoisted
                        float px = cumulSp * lp.energyScaleFact;
                        float fx = floorf(px), ax = px - fx;
                        int x0 = (int)fx, x1 = x0 + 1;
                        if (!(px >= 0.0f)) { x0 = 0; x1 = 0; ax = 0.0f; }
                        x0 = x0 > lut.nSamples - 1 ? lut.nSamples - 1 : x0; x1 = x1 > lut.nSamples - 1 ? lut.nSamples - 1 : x1;
                        float r0 = LDS_LUT ? lerpW(ax, sRow0[x0], sRow0[x1]) : lerpW(ax, gRow0[x0], gRow0[x1]);
                        float r1 = LDS_LUT ? lerpW(ax, sRow1[x0], sRow1[x1]) : lerpW(ax, gRow1[x0], gRow1[x1]);
                        cumulDose = lerpW(eay, r0, r1);
                    }
                    if (cumulSp > cutDepth || stepNo == afterLast) { beamLive = false; afterLast = stepNo; }
                    // stepTab[2k+1] = stepVol(k) (fill_idd_and_sigma_params.cu:72)
                    const float stepVol = stepTab[2 * stepNo + 1];
                    const float mass = fc.doseToWater ? (cumulSp - cumulSpOld) * stepVol : density * stepVol;
                    // (the dose value feeds no threshold other than res > 0, which a reciprocal cannot change: hardware reciprocal, <= 1 ulp)
                    if (!NUC || !fc.nuclearCorr) {
                        if (mass > 1e-2f) res = rayWeight * (cumulDose - cumulDoseOld) * __builtin_amdgcn_rcpf(mass);
                    } else if (mass > 1e-2f) {                       // :320-331: the primary keeps (1 - nucWeight), the halo gets nucWeight
                        const float nucWeight = sample2dClamp(lut.nucWeight, lut.nSamples, lut.nEnergies,
                                                              0.5f * (cumulSp + cumulSpOld) * lp.energyScaleFact, lp.energyIdx);
                        res = (1.0f - nucWeight) * rayWeight * (cumulDose - cumulDoseOld) * __builtin_amdgcn_rcpf(mass);
                        nucRes = nucWeight * nucRayWeight * (cumulDose - cumulDoseOld) / (mass * fc.spotDist * fc.spotDist);
                    }
                    cumulSpOld = cumulSp;
                    cumulDoseOld = cumulDose;
                }
                if (!beamLive || (int)stepNo < (firstIn - 1)) { res = 0.0f; nucRes = 0.0f; }
                (bevIdd + layerOff + (size_t)stepNo * memStep)[rayOff] = res;
                doseMask[j] = __ballot(res > 0.0f);
            }
            if ((tid & (kWave - 1)) == 0) {
#pragma unroll
                for (int j = 0; j < kFillBatch; ++j) sDoseMask[buf][j][wave] = doseMask[j];
            }
            ldsBarrier();                                            // the only barrier of a batch (sDoseMask is double-buffered)
            // rectangle of the tile's rays that carry dose at step j, as minima of (x, y, -x, -y): lanes 0..3 of the step's group,
            // one component each, from the four waves' ballots (wave w holds rows 2w, 2w+1 of the tile: low / high 32 bits)
            const int j = tid >> 5, l = tid & 31;
            if (l < 4 && step0 + j < pAfterLast) {
                unsigned int colMask = 0u, rowMask = 0u;             // columns / rows of the tile with dose
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned long long dm = sDoseMask[buf][j][w];
                    const unsigned int lo = (unsigned int)dm, hi = (unsigned int)(dm >> 32);
                    colMask |= lo | hi;
                    rowMask |= (lo ? 1u : 0u) << (2 * w) | (hi ? 1u : 0u) << (2 * w + 1);
                }
                if (colMask) {
                    const int x0t = tileX * kSuperpTileX, y0t = tileY * kSuperpTileY;
                    const int v = l == 0 ? x0t + __builtin_ctz(colMask) : l == 1 ? y0t + __builtin_ctz(rowMask)
                                : l == 2 ? -(x0t + 31 - __builtin_clz(colMask)) : -(y0t + 31 - __builtin_clz(rowMask));
                    atomicMin(&active[((size_t)layer * fc.S + step0 + j) * 4 + l], v);
                    actUni = min(actUni, v);
                }
            }
        }
        if ((tid & 31) < 4 && actUni != 0x7fffffff) atomicMin(&st->actUnion[tid & 3], actUni);
        if (NUC && nucIdx >= 0 && pFirst < pAfterLast) nuc.idd[(size_t)layer * fc.nucW * fc.nucH + nucIdx] = nucRes;   // value of the last step (:367-373)
    }
    if (dbg && tid == 0) {
        long long* q = dbg + 4 * (size_t)item;
        q[0] = dbgT0; q[1] = (long long)__builtin_amdgcn_s_memtime();
        q[2] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        q[3] = ((long long)item << 8) | (long long)(role << 4) | 0;
        (void)pAfterLast;
    }
}

// The two boxes of a transfer (kernel_wrapper.cu:1185-1213): bbox = the reference's minIdx / maxIdx from the eight corners of the
// padded BEV cube (W x H rays, slices [first, calcPassive)); tbox = the voxels the transfer can actually change: the image of
// the block of the slab that can be non-zero — pixels [bevLo, bevHi], slices [slabLo, slabHi) — grown by the interpolation reach,
// clipped by the voxels the reference's launch visits.
__device__ inline void transferBoxes(const FromFan& rayIdxToDoseIdx, int W, int H, int first, int calcPassive, const int bevLo[2], const int bevHi[2],
                                     int slabLo, int slabHi, int doseNx, int doseNy, int doseNz, int bboxMin[3], int bboxMax[3], int tboxMin[3], int tboxMax[3]) {
    Vec3 maxP = v3(-1.0f, -1.0f, -1.0f), minP = v3(100000.0f, 100000.0f, 100000.0f);
    float xVals[2] = { -(float)kMaxSuperpR, (float)(W + kMaxSuperpR - 1) };
    float yVals[2] = { -(float)kMaxSuperpR, (float)(H + kMaxSuperpR - 1) };
    float zVals[2] = { (float)first, (float)(calcPassive - 1) };
    for (int zi = 0; zi < 2; ++zi) for (int yi = 0; yi < 2; ++yi) for (int xi = 0; xi < 2; ++xi) {
        Vec3 p = transformPoint(rayIdxToDoseIdx, v3(xVals[xi], yVals[yi], zVals[zi]));
        if (p.x > maxP.x) maxP.x = p.x; if (p.y > maxP.y) maxP.y = p.y; if (p.z > maxP.z) maxP.z = p.z;
        if (p.x < minP.x) minP.x = p.x; if (p.y < minP.y) minP.y = p.y; if (p.z < minP.z) minP.z = p.z;
    }
    int t;
    t = (((int)floorf(minP.x)) / 32) * 32; bboxMin[0] = t > 0 ? t : 0;
    t = (int)floorf(minP.y); bboxMin[1] = t > 0 ? t : 0;
    t = (int)floorf(minP.z); bboxMin[2] = t > 0 ? t : 0;
    t = (int)ceilf(maxP.x); bboxMax[0] = t < doseNx - 1 ? t : doseNx - 1;
    t = (int)ceilf(maxP.y); bboxMax[1] = t < doseNy - 1 ? t : doseNy - 1;
    t = (int)ceilf(maxP.z); bboxMax[2] = t < doseNz - 1 ? t : doseNz - 1;
    // The voxels primTransfDiv visits (kernel_wrapper.cu:69-97, launch :1209-1214): its grid starts at minIdx and is rounded up
    // to whole 32 x 8 blocks, clipped by the dose dimensions only — x and y run PAST maxIdx up to the block edge — while z
    // stops at maxIdx.z. Voxels between maxIdx and the block edge do receive dose when the interpolated BEV value there
    // is non-zero (one BEV step beyond the last slice still interpolates against it), so the coverage is kept exactly.
    const int covMax[3] = { min(bboxMin[0] + roundToI(bboxMax[0] - bboxMin[0] + 1, 32) - 1, doseNx - 1),
                            min(bboxMin[1] + roundToI(bboxMax[1] - bboxMin[1] + 1, 8) - 1, doseNy - 1), bboxMax[2] };
    // the BEV dose is exactly zero outside the padded rectangle [bevLo, bevHi] and outside the slices [slabLo, slabHi):
    // the image of that block, grown by the interpolation reach (one pixel / one step on every side), bounds the voxels
    // the transfer can change
    float txVals[2] = { (float)(bevLo[0] - 32 - 1), (float)(bevHi[0] - 32 + 1) };
    float tyVals[2] = { (float)(bevLo[1] - 32 - 1), (float)(bevHi[1] - 32 + 1) };
    float tzVals[2] = { (float)(slabLo - 1), (float)slabHi };
    maxP = v3(-1.0f, -1.0f, -1.0f); minP = v3(100000.0f, 100000.0f, 100000.0f);
    for (int zi = 0; zi < 2; ++zi) for (int yi = 0; yi < 2; ++yi) for (int xi = 0; xi < 2; ++xi) {
        Vec3 p = transformPoint(rayIdxToDoseIdx, v3(txVals[xi], tyVals[yi], tzVals[zi]));
        if (p.x > maxP.x) maxP.x = p.x; if (p.y > maxP.y) maxP.y = p.y; if (p.z > maxP.z) maxP.z = p.z;
        if (p.x < minP.x) minP.x = p.x; if (p.y < minP.y) minP.y = p.y; if (p.z < minP.z) minP.z = p.z;
    }
    const int lo[3] = { (((int)floorf(minP.x) - 1) / 32) * 32,   // (aligned like the reference's box, :1207)
                        (int)floorf(minP.y) - 1, (int)floorf(minP.z) - 1 };
    const int hi[3] = { (int)ceilf(maxP.x) + 1, (int)ceilf(maxP.y) + 1, (int)ceilf(maxP.z) + 1 };
    for (int i = 0; i < 3; ++i) {
        tboxMin[i] = lo[i] > bboxMin[i] ? lo[i] : bboxMin[i];
        tboxMax[i] = hi[i] < covMax[i] ? hi[i] : covMax[i];
    }
}

// ------------------------------------------------------------------------------------------------
// K6: superposition plan = host batching of radii (kernel_wrapper.cu:965-976) + beamFirstCalculatedPassive
// (:955-957) + transfer bounding box and shift (:1185-1213), all on the device.
struct KsPlanArgs {
    FieldState* stGlobal; LayerPlan* layers; FromFan rayIdxToDoseIdx; TransferParams tp0;
    int doseNx, doseNy, doseNz, G, Gs;
    FieldState* hostMirror; FieldState* stNuc;
    const unsigned int* sigMin; const unsigned int* sigMax;
    int uniformEligible, sweepMaxR, Gb;
};
// The batching rule of one layer (kernel_wrapper.cu:966-976): batch radius per radius class from the layer's class histogram; returns
// the largest class present. Used by the plan (which records the result) and by k_superpose_sweep's blocks when they plan for themselves.
__device__ inline int batchRadii(const int (&hist)[kMaxSuperpR + 2], int (&effRad)[kMaxSuperpR + 2]) {
    int layerMax = 0;
#pragma unroll
    for (int i = 0; i < kMaxSuperpR + 2; ++i) { if (hist[i] > 0) layerMax = i; effRad[i] = i; }
    // tiles at steps >= layerFirstPassive are not classified by the reference; they can only be radius 0
    if (layerMax <= kMaxSuperpR) {
        int rec = layerMax, batched = 0;
#pragma unroll
        for (int rad = kMaxSuperpR; rad > 0; --rad) {
            if (rad <= layerMax) {
                batched += hist[rad];
                effRad[rad] = rec;
                if (batched >= kMinTilesInBatch) { rec = rad - 1; batched = 0; }
            }
        }
    }
    return layerMax;
}
// The block's LDS (k_ks_plan: static; as block 0 of k_superpose_sweep's launch: the front of that kernel's dynamic LDS, so that its
// blocks' footprint — two per CU — does not grow)
struct KsPlanLds {
    FieldState st;
    unsigned long long live;
    int maxPassive, maxRad, sliceDiffers;
    int group[32], groupSw[16], bigLo[16], bigHi[16];
    int area[kKsMaxOrder];
};
// One block of nT threads (a launch of its own, k_ks_plan, or block 0 of k_superpose_sweep's launch — rtd_sweep.hpp).
__device__ inline void ksPlanBody(const KsPlanArgs& ka, const FieldConst& fc, const int tid, const int nT, KsPlanLds& L_) {
    FieldState* stGlobal = ka.stGlobal; LayerPlan* layers = ka.layers;
    const FromFan& rayIdxToDoseIdx = ka.rayIdxToDoseIdx; const TransferParams& tp0 = ka.tp0;
    const int doseNx = ka.doseNx, doseNy = ka.doseNy, doseNz = ka.doseNz, G = ka.G, Gs = ka.Gs;
    FieldState* __restrict__ hostMirror = ka.hostMirror; FieldState* __restrict__ stNuc = ka.stNuc;
    const unsigned int* __restrict__ sigMin = ka.sigMin; const unsigned int* __restrict__ sigMax = ka.sigMax;   // (restrict: the loads of the uniformity test stay in flight together)
    const int uniformEligible = ka.uniformEligible, sweepMaxR = ka.sweepMaxR, Gb = ka.Gb;
    // The state record is completed in LDS and then written out — to device memory and to its pinned host mirror — by all threads,
    // one dword each per trip: this one-block launch sits on the field's critical path, and both a load of the record behind a
    // store to it and a serial copy over PCIe by one thread cost microseconds each.
    FieldState& sSt = L_.st;
    int& sMaxPassive = L_.maxPassive; int& sMaxRad = L_.maxRad; int& sSliceDiffers = L_.sliceDiffers;
    unsigned long long& sLive = L_.live;
    int (&sGroup)[32] = L_.group; int (&sGroupSw)[16] = L_.groupSw; int (&sBigLo)[16] = L_.bigLo; int (&sBigHi)[16] = L_.bigHi;
    {
        const unsigned int* src = reinterpret_cast<const unsigned int*>(stGlobal);
        unsigned int* dst = reinterpret_cast<unsigned int*>(&sSt);
        for (unsigned int i = tid; i < sizeof(FieldState) / 4; i += nT) dst[i] = src[i];
    }
    if (tid == 0) { sMaxPassive = 0; sMaxRad = 0; sLive = 0ull; sSliceDiffers = 0; }
    if (tid < 32) sGroup[tid] = 0;
    if (tid < 16) { sGroupSw[tid] = 0; sBigLo[tid] = 0x7fffffff; sBigHi[tid] = 0; }
    __syncthreads();
    FieldState* st = &sSt;
    const int first = st->beamFirstInside;
    const int au[4] = { st->actUnion[0], st->actUnion[1], st->actUnion[2], st->actUnion[3] };
    for (int l = tid; l < fc.L; l += nT) {
        LayerPlan& p = layers[l];
        int hist[kMaxSuperpR + 2], effRad[kMaxSuperpR + 2];          // one round trip for the histogram, one for the result
#pragma unroll
        for (int i = 0; i < kMaxSuperpR + 2; ++i) hist[i] = p.hist[i];
        const int lfp = p.layerFirstPassive;
        const int layerMax = batchRadii(hist, effRad);
        if (hist[kMaxSuperpR + 1] > 0) atomicOr(&st->errorFlags, kErrRadiusOverflow);
#pragma unroll
        for (int i = 0; i < kMaxSuperpR + 2; ++i) p.effRad[i] = effRad[i];
        atomicMax(&sMaxRad, layerMax);
        atomicMax(&sMaxPassive, lfp);
        atomicMax(&sGroup[l % G], lfp);
        atomicMax(&sGroupSw[l % Gs], lfp);
        if (lfp > first) atomicAdd(&sLive, (unsigned long long)(lfp - first));
        // the steps of the layer with a tile whose batch radius the sweep's first launch does not take (k_fill recorded the steps of
        // every radius class; the batch radius of a class is known only here)
        if (sweepMaxR >= 0 && layerMax <= kMaxSuperpR && layerMax > sweepMaxR) {
            int lo = 0x7fffffff, hi = -1;                            // (one more round trip — unconditional loads, all in flight — only in a field with such radii)
#pragma unroll
            for (int i = 1; i <= kMaxSuperpR; ++i) {
                const int a = p.classLo[i], b = p.classHi[i];
                const bool big = hist[i] > 0 && effRad[i] > sweepMaxR;
                lo = big ? min(lo, a) : lo; hi = big ? max(hi, b) : hi;
            }
            if (hi >= lo) { atomicMin(&sBigLo[l % Gb], lo); atomicMax(&sBigHi[l % Gb], hi + 1); }
        }
    }
    __syncthreads();
    // Uniform-sigma field (water)? No tile saw two sigma^2 (k_fill) and every depositing (layer, step) has one over all its tiles.
    // A heterogeneous field leaves here at the first test.
    const bool maybeUniform = uniformEligible && !st->nonUniform && !st->errorFlags;
    if (maybeUniform) {
        // (entries of steps outside [first, layerFirstPassive) still hold the reset values (+inf, 0) or a uniform tile's value: the
        //  test "+inf or equal" needs no step range, and with four pairs of loads in flight the 2 x L x S words cost ~5 us)
        int differs = 0;
        const int n = fc.L * fc.S;
        for (int i0 = tid; i0 < n; i0 += 4 * nT) {
            unsigned int a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                            // (unconditional loads from a clamped index: all eight in flight together)
                const int ii = min(i0 + u * nT, n - 1);
                a[u] = sigMin[ii]; b[u] = sigMax[ii];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (a[u] != 0x7f800000u && a[u] != b[u]) differs = 1;   // (+inf: no live ray; a repeated last entry changes nothing)
        }
        if (differs) atomicOr(&sSliceDiffers, 1);
    }
    __syncthreads();
    {   // Output tiles of the superposition ranked by the number of dose-carrying rays within reach: the work items of the
        // busiest tiles are dispatched first, the items of margin tiles (short) last, so the kernel does not end on a few
        // long items. One tile per thread, stable rank by counting.
        int (&sArea)[kKsMaxOrder] = L_.area;
        const int nTX = (fc.bevW + kKsTileX - 1) / kKsTileX, nTY = (fc.bevH + kKsTileY - 1) / kKsTileY, n = nTX * nTY;
        if (n <= kKsMaxOrder) {                                      // (kKsMaxOrder <= nT)
            const int rr = min(sMaxRad, kMaxSuperpR), t = tid;
            int area = 0;
            if (t < n) {
                const int ox0 = (t % nTX) * kKsTileX, oy0 = (t / nTX) * kKsTileY;
                const int w = min(ox0 + 31 + rr, -au[2]) - max(ox0 - 32 - rr, au[0]) + 1;
                const int h = min(oy0 - 1 + rr, -au[3]) - max(oy0 - 32 - rr, au[1]) + 1;
                area = (w > 0 && h > 0) ? w * h : 0;
                sArea[t] = area;
            }
            __syncthreads();
            if (t < n) {
                int rank = 0;
                for (int u = 0; u < n; ++u) { const int au = sArea[u]; rank += (au > area || (au == area && u < t)) ? 1 : 0; }
                st->tileOrder[rank] = (unsigned char)t;
            }
        }
    }
    if (tid == 0) {
        // (everything is computed in registers from values read once, and stored at the end: a load of *st behind a store to it is
        //  a full memory round trip, and this thread is the critical path of the launch)
        const int calcPassive = sMaxPassive;
        const int rr = min(sMaxRad, kMaxSuperpR);
        // a source at ray (x, y) reaches padded BEV pixels (x+32 +- r, y+32 +- r), r <= the largest batch radius
        const int bevLo[2] = { au[0] + 32 - rr, au[1] + 32 - rr }, bevHi[2] = { -au[2] + 32 + rr, -au[3] + 32 + rr };
        TransferParams tp = tp0;
        tp.globalOffset.z = tp0.globalOffset.z + (-(float)first);   // invertAndShift(..., -beamFirstInside) :1213
        int bboxMin[3] = {0, 0, 0}, bboxMax[3] = {0, 0, 0}, tboxMin[3] = {0, 0, 0}, tboxMax[3] = {-1, -1, -1};
        if (calcPassive > first)
            transferBoxes(rayIdxToDoseIdx, fc.W, fc.H, first, calcPassive, bevLo, bevHi, first, calcPassive, doseNx, doseNy, doseNz,
                          bboxMin, bboxMax, tboxMin, tboxMax);
        st->firstCalculatedPassive = calcPassive;
        st->uniformField = (maybeUniform && !sSliceDiffers) ? 1 : 0;
        st->maxRadius = sMaxRad;
        for (int gI = 0; gI < 32; ++gI) st->groupPassive[gI] = sGroup[gI];
        for (int gI = 0; gI < 16; ++gI) { st->swGroupPassive[gI] = sGroupSw[gI]; st->swBigFirst[gI] = sBigLo[gI]; st->swBigPassive[gI] = sBigHi[gI]; }
        st->bevLo[0] = bevLo[0]; st->bevLo[1] = bevLo[1]; st->bevHi[0] = bevHi[0]; st->bevHi[1] = bevHi[1];
        st->liveSteps = (long long)sLive;
        st->packX0 = 0; st->packY0 = 0; st->packW = fc.bevW; st->packH = fc.bevH; st->slabFirst = first;
        st->transfer = tp;
        for (int i = 0; i < 3; ++i) { st->bboxMin[i] = bboxMin[i]; st->bboxMax[i] = bboxMax[i]; st->tboxMin[i] = tboxMin[i]; st->tboxMax[i] = tboxMax[i]; }
    }
    // The state record is final here (the kernels after this one only read it): all threads write it to device memory and
    // mirror it into pinned host memory, so rtd_field_finish needs no device-to-host copy (a copy on a second stream stalled
    // the compute queue for ~37 us per field, measured). Kernel completion makes both copies visible; no fence is needed.
    __syncthreads();
    {
        const unsigned int* src = reinterpret_cast<const unsigned int*>(&sSt);
        unsigned int* dst = reinterpret_cast<unsigned int*>(stGlobal);
        volatile unsigned int* mir = reinterpret_cast<volatile unsigned int*>(hostMirror);
        for (unsigned int i = tid; i < sizeof(FieldState) / 4; i += nT) {
            const unsigned int v = src[i];
            dst[i] = v;
            if (hostMirror) mir[i] = v;
        }
        // NUCLEAR_CORR: a radius overflow of the primary field stops the halo's transfer as well
        if (stNuc && tid == 0 && sSt.errorFlags) stNuc->errorFlags = sSt.errorFlags;
    }
}
__global__ __launch_bounds__(1024) void k_ks_plan(KsPlanArgs ka, FieldConst fc) {
    __shared__ KsPlanLds lds;
    ksPlanBody(ka, fc, threadIdx.x, blockDim.x, lds);
}

// ------------------------------------------------------------------------------------------------
// NUCLEAR_CORR halo (default off): what the reference's nuclear launches do given its fill (see NucFill).
// Per layer the reference classifies the tiles of the nuclear arrays for the steps [entry, layerFirstPassive)
// (kernel_wrapper.cu:978-997) and superposes them (:1058-1091); only plane 0 of those arrays ever holds anything but the
// initial (0, inf), so the halo cube receives dose in slice 0 only, and only when the beam's entry step is 0.
// k_nuc_plan   one block: radius class of every (layer, tile) of plane 0, the batching rule per layer, the state record of the
//              one-slice halo slab (boxes, transfer parameters). A radius overflow is reported in the PRIMARY state (it runs
//              before k_ks_plan), like the reference's throw at :984.
// k_nuc_superpose   one thread per pixel of the padded halo slice: the same patches as kernelSuperposition (:432-489), gathered.
__global__ __launch_bounds__(256) void k_nuc_plan(FieldState* stPrim, FieldState* stNuc, const LayerPlan* __restrict__ layers,
                                                  const float* __restrict__ nucRs, int* __restrict__ nucEffT, FieldConst fc,
                                                  FromFan nucIdxToDoseIdx, TransferParams tp0, int doseNx, int doseNy, int doseNz) {
    __shared__ int sAny, sCalc;
    if (threadIdx.x == 0) { sAny = 0; sCalc = 0; }
    __syncthreads();
    const int first = stPrim->beamFirstInside;
    const int tX = fc.nucW / kSuperpTileX, tY = fc.nucH / kSuperpTileY, nT = tX * tY;
    const size_t nucR = (size_t)fc.nucW * fc.nucH;
    for (int l = threadIdx.x; l < fc.L; l += blockDim.x) {
        const int lfp = layers[l].layerFirstPassive;
        atomicMax(&sCalc, lfp);
        for (int t = 0; t < nT; ++t) nucEffT[l * nT + t] = -1;
        if (first != 0 || lfp <= 0) continue;                        // plane 0 is not among the steps [first, layerFirstPassive)
        int hist[kMaxSuperpR + 2];
        for (int i = 0; i < kMaxSuperpR + 2; ++i) hist[i] = 0;
        for (int t = 0; t < nT; ++t) {                               // tileRadCalc (kernel_wrapper.cuh:256-313) on plane 0
            const float* base = nucRs + (size_t)l * nucR + (size_t)(t / tX) * kSuperpTileY * fc.nucW + (t % tX) * kSuperpTileX;
            float m = base[0];
            for (int r = 0; r < kSuperpTileY; ++r) for (int c = 0; c < kSuperpTileX; ++c) { const float v = base[r * fc.nucW + c]; m = v < m ? v : m; }
            int rad = f2iSat(fc.ksSigmaCutoff / (sqrtf(2.0f) * m) + 0.5f);
            rad = rad > kMaxSuperpR + 1 ? kMaxSuperpR + 1 : (rad < 0 ? 0 : rad);
            hist[rad] += 1;
            nucEffT[l * nT + t] = rad;
        }
        if (hist[kMaxSuperpR + 1] > 0) { atomicOr(&stPrim->errorFlags, kErrRadiusOverflow); continue; }   // :984
        int layerMax = 0, eff[kMaxSuperpR + 2];
        for (int i = 0; i < kMaxSuperpR + 2; ++i) { if (hist[i] > 0) layerMax = i; eff[i] = i; }
        int rec = layerMax, batched = 0;                             // batching rule, :986-996
        for (int rad = layerMax; rad > 0; --rad) {
            batched += hist[rad];
            eff[rad] = rec;
            if (batched >= kMinTilesInBatch) { rec = rad - 1; batched = 0; }
        }
        for (int t = 0; t < nT; ++t) nucEffT[l * nT + t] = eff[nucEffT[l * nT + t]];
        atomicOr(&sAny, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        FieldState s;
        for (unsigned int i = 0; i < sizeof(FieldState) / 4; ++i) reinterpret_cast<unsigned int*>(&s)[i] = 0u;
        const int bevW = fc.nucW + 2 * kMaxSuperpR, bevH = fc.nucH + 2 * kMaxSuperpR;
        s.beamFirstInside = 0;
        s.firstCalculatedPassive = sAny ? 1 : 0;                     // the one slice that can hold dose
        s.bevLo[0] = 0; s.bevLo[1] = 0; s.bevHi[0] = bevW - 1; s.bevHi[1] = bevH - 1;
        s.packX0 = 0; s.packY0 = 0; s.packW = bevW; s.packH = bevH; s.slabFirst = 0;
        s.transfer = tp0;                                            // (shift by -beamFirstInside = 0, :1245)
        for (int i = 0; i < 3; ++i) { s.tboxMin[i] = 0; s.tboxMax[i] = -1; }
        if (sAny) transferBoxes(nucIdxToDoseIdx, fc.nucW, fc.nucH, 0, sCalc, s.bevLo, s.bevHi, 0, 1, doseNx, doseNy, doseNz,
                                s.bboxMin, s.bboxMax, s.tboxMin, s.tboxMax);
        *stNuc = s;
    }
}

__global__ __launch_bounds__(256) void k_nuc_superpose(const float* __restrict__ nucIdd, const float* __restrict__ nucRs, const int* __restrict__ nucEffT,
                                                       const FieldState* __restrict__ stNuc, FieldConst fc, float* __restrict__ bevNuc) {
    const int bevW = fc.nucW + 2 * kMaxSuperpR, bevH = fc.nucH + 2 * kMaxSuperpR;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= bevW * bevH) return;
    float acc = 0.0f;
    if (stNuc->firstCalculatedPassive > 0 && !stNuc->errorFlags) {
        const int px = pix % bevW, py = pix / bevW;
        const int tX = fc.nucW / kSuperpTileX, nT = tX * (fc.nucH / kSuperpTileY);
        const size_t nucR = (size_t)fc.nucW * fc.nucH;
        for (int l = 0; l < fc.L; ++l)
            for (int sy = max(py - 32 - kMaxSuperpR, 0); sy <= min(py - 32 + kMaxSuperpR, fc.nucH - 1); ++sy)
                for (int sx = max(px - 32 - kMaxSuperpR, 0); sx <= min(px - 32 + kMaxSuperpR, fc.nucW - 1); ++sx) {
                    const int rho = nucEffT[l * nT + (sy / kSuperpTileY) * tX + sx / kSuperpTileX];
                    const int dx = abs(px - 32 - sx), dy = abs(py - 32 - sy);
                    if (rho < 0 || dx > rho || dy > rho) continue;
                    const float dose = nucIdd[(size_t)l * nucR + (size_t)sy * fc.nucW + sx];
                    if (!(dose > 0.0f)) continue;
                    const float rs = nucRs[(size_t)l * nucR + (size_t)sy * fc.nucW + sx];
                    // erfDiffs (kernel_wrapper.cuh:459-467)
                    const float ex = 0.5f * (erff(rs * ((float)dx + 0.5f)) - erff(rs * ((float)dx - 0.5f)));
                    const float ey = 0.5f * (erff(rs * ((float)dy + 0.5f)) - erff(rs * ((float)dy - 0.5f)));
                    acc += dose * ey * ex;
                }
    }
    bevNuc[pix] = acc;
}

// ------------------------------------------------------------------------------------------------
// K7: output-stationary kernel superposition on the matrix cores, one autonomous WAVE per work item.
//
// Same arithmetic as kernelSuperposition<rad> (kernel_wrapper.cuh:432-489): every source voxel s adds the
// separable patch  dose_s * e_s[|dy|] * e_s[|dx|],  |dy|,|dx| <= rho_s  (rho_s = batch radius of its 32x8 tile,
// e_s = erf-difference weights of ITS OWN 1/sigma). A patch is a rank-1 update, so a wave that OWNS a 32x64
// tile of the padded BEV slice at step k accumulates  D += A * B  with
//     A[r][s] = dose_s * m_s[|r - y_s|],   B[s][c] = m_s[|c - x_s|]   (m_s = one-sided weight table of source s, 0 beyond rho_s)
// on v_mfma_f32_16x16x4_f32 (exact f32 FMA chain at twice the f32 vector FMA rate; the 8 accumulator tiles have static
// register indices while the operands are data, which a per-source-radius VALU loop cannot have).
// Work item = (output tile, step k, layer group g): the wave walks the layers l = g, g+G, ... and, per layer,
// the source window in reach in chunks of <= 64 sources; per chunk it computes which of its 8 MFMA tiles every source
// reaches (own batch radius), builds the weight tables of the chunk into its private LDS slice (one source per lane,
// the erfDiffs weights of kernel_wrapper.cuh:459-467), then issues one MFMA per (source quad, 16x16 tile) pair in the
// quad's reach mask. No block barrier, no float atomics (the reference's flush, kernel_wrapper.cuh:486), no zero-fill
// pass (kernel_wrapper.cu:824-827): the groups' accumulators are added in a fixed binary tree inside this launch (epilogue),
// so the BEV dose is bitwise reproducible.
typedef float f32x4 __attribute__((ext_vector_type(4)));
// kKsSplit (template parameter of the kernel) = waves per work item: its source chunks are dealt round-robin to them and the
// accumulators are summed through LDS at the end, in fixed order. 1 (single-wave blocks) when there are enough items to fill the
// chip (C3: 2 = no gain, 4 = slower); 2 or 4 for fields with few layers, where the items are too few and too long (C1, one
// layer: 1872 live items for 7168 wave slots) — chosen on the host from the item count.
constexpr int kKsWaveLds = 1200;              // floats of LDS per wave (4.7 KiB): CS source blocks (dose, guard address, T entries); with the reach table
                                              // 5 KiB per block, so LDS admits 31 blocks per CU and the 72 VGPRs 7 waves per SIMD
constexpr int kKsReachTiles = 80;            // 32x8 source tiles within +-32 of a 64x32 output tile: <= 5 x 13
constexpr int kKsMaxGroups = 32;              // upper bound of layer groups (= partial BEV buffers)

__device__ inline int clampI(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int kKsSplit>
__global__ __launch_bounds__(64 * kKsSplit, 7) void k_superpose_mfma(const float* __restrict__ bevIdd, const float* __restrict__ bevRSigmaEff,
                                                            float* __restrict__ bevPart, const unsigned char* __restrict__ tileRad,
                                                            const LayerPlan* __restrict__ layers, const FieldState* __restrict__ st,
                                                            FieldConst fc, int nTX, int nTY, int G, const int* __restrict__ active,
                                                            float* __restrict__ bevDose, int* __restrict__ nodeCount, int sweepMaxR) {
    constexpr int kSlice = kKsWaveLds + kKsReachTiles;
    static_assert(kKsSplit == 1 || kKsSplit * kSlice >= 2048, "the accumulator exchange needs 2048 floats of LDS");
    __shared__ __attribute__((aligned(16))) float ldsAll[kKsSplit * kSlice];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave of the item (wave-uniform)
    float* lds = ldsAll + wv * kSlice;                             // this wave's private slice
    int* effT = reinterpret_cast<int*>(lds + kKsWaveLds);          // batch radius of every source tile in reach (-1: none)
    const int lane = threadIdx.x & 63;
    // One wave per block: a heavy item never keeps three finished neighbours' LDS and wave slots occupied.
    // decode the work item (wave-uniform): fastest index = layer group, then step, then output tile
    int item = blockIdx.x;
    const int gi = item % G; item /= G;
    const int ki = item % fc.S; item /= fc.S;
    const int k = fc.S - 1 - ki;                                      // within a tile the deepest steps (largest radii) go first
    // the group index is rotated with the step: with all CUs busy block b tends to land on CU b % nCU, and a fixed position of the
    // groups that hold two layers (L > G) would put all the double-work items on the same CUs when G divides the CU count
    // (measured: G = 16 on 256 CUs 0.88 ms against 0.60 ms)
    const int g = (gi + ki) % G;
    const int tile = nTX * nTY <= kKsMaxOrder ? st->tileOrder[item] : item;   // busiest tiles first (k_ks_plan)
    const int tX = tile % nTX, tY = tile / nTX;
    const int first = st->beamFirstInside, calcPassive = st->firstCalculatedPassive;
    if (st->errorFlags) return;                                      // radius overflow: the reference throws before any superposition (kernel_wrapper.cu:965)
    if (st->uniformField) return;                                    // one sigma per slice: the separable kernel (rtd_uniform.hpp) has written the BEV dose
    if (st->maxRadius <= sweepMaxR) return;                          // every batch radius within k_superpose_sweep's reach: it writes the BEV dose
    if (k < 0 || k < first || k >= calcPassive) return;
    const int li = lane & 15, kq = lane >> 4;                         // MFMA 16x16x4: A[i=li][k=kq], B[k=kq][j=li]
    const int ox0 = tX * kKsTileX, oy0 = tY * kKsTileY;               // padded BEV coordinates of the owned tile
    // A tile outside the rectangle that any patch of the field can reach receives nothing: one item per (tile, slice) writes its
    // zeros into the BEV dose (the transfer interpolates against the pixels next to the rectangle).
    if (ox0 > st->bevHi[0] || ox0 + kKsTileX - 1 < st->bevLo[0] || oy0 > st->bevHi[1] || oy0 + kKsTileY - 1 < st->bevLo[1]) {
        if (gi == 0 && wv == 0) {
            float* dst = bevDose + (size_t)k * fc.bevW * fc.bevH;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                const int oy = oy0 + 16 * (e >> 4) + 4 * kq + (e & 3), ox = ox0 + 16 * ((e >> 2) & 3) + li;
                if (oy < fc.bevH && ox < fc.bevW) dst[(size_t)oy * fc.bevW + ox] = 0.0f;
            }
        }
        return;
    }
    if (k >= st->groupPassive[g]) return;                             // no layer of this group deposits at k
    const int W = fc.W, H = fc.H;
    const size_t memStep = (size_t)W * H;
    const int nTiles = fc.tilesX * fc.tilesY;

    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    for (int layer = g; layer < fc.L; layer += G) {
        if (k >= layers[layer].layerFirstPassive) continue;          // nothing deposited by this layer at this step
        const int* eff = layers[layer].effRad;
        const unsigned char* tr = tileRad + ((size_t)layer * fc.S + k) * nTiles;
        // ---- reach: largest batch radius among the 32x8 source tiles whose patches can touch the owned tile ----
        int rho = -1;
        // source-coordinate rectangle of the owned tile: [ox0-32, ox0+31] x [oy0-32, oy0-1]
        const int tx0 = clampI((ox0 - 32 - kMaxSuperpR) >> 5, 0, fc.tilesX - 1), tx1 = clampI((ox0 + 31 + kMaxSuperpR) >> 5, 0, fc.tilesX - 1);
        const int ty0 = clampI((oy0 - 32 - kMaxSuperpR) >> 3, 0, fc.tilesY - 1), ty1 = clampI((oy0 - 1 + kMaxSuperpR) >> 3, 0, fc.tilesY - 1);
        const int ntx = tx1 - tx0 + 1, nt = ntx * (ty1 - ty0 + 1);   // <= kKsReachTiles
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < nt; t += kWave) {
            const int tx = tx0 + t % ntx, ty = ty0 + t / ntx;
            const int own = tr[ty * fc.tilesX + tx];
            const int r = own <= kMaxSuperpR ? eff[own] : -1;        // unclassified (0xFF) or overflow (reported via errorFlags)
            effT[t] = r;
            const int gx = max(max(tx * 32 - (ox0 + 31), (ox0 - 32) - (tx * 32 + 31)), 0);
            const int gy = max(max(ty * 8 - (oy0 - 1), (oy0 - 32) - (ty * 8 + 7)), 0);
            if (max(gx, gy) <= r) rho = max(rho, r);
        }
        rho = waveMaxI(rho);
        if (rho < 0) continue;                                       // wave-uniform
        const int Tm = rho + 1, T = Tm + 1;                          // one-sided table m[u], u = |d| in [0, Tm]; m[Tm] = 0 (weights are even in d)
        // source window = reach of the tile, clipped to the ray grid and to the rectangle of rays that carry dose at this
        // (layer, step) (recorded by k_fill): margins of dead rays are never scanned
        const int* act = active + ((size_t)layer * fc.S + k) * 4;
        const int cx0 = max(max(ox0 - 32 - rho, 0), act[0]), cx1 = min(min(ox0 + 31 + rho + 1, W), -act[2] + 1);
        const int ry0 = max(max(oy0 - 32 - rho, 0), act[1]), ry1 = min(min(oy0 - 1 + rho + 1, H), -act[3] + 1);
        if (cx1 <= cx0 || ry1 <= ry0) continue;
        // LDS per source: (dose, address of its table's zero guard) in front of its T table entries, 8-byte aligned
        const int TS = (T + 3) & ~1;                                 // floats per source
        const int CS = min(kWave, (kKsWaveLds / TS) & ~3);           // sources per chunk (whole quads)
        const size_t sliceOff = (size_t)layer * memStep * fc.S + (size_t)k * memStep;

        // The window's sources are walked row-major in chunks of CS (rows padded to whole quads), so a chunk may
        // span several source rows and every lane builds one table.
        const int nCols = ((cx1 - cx0 + 3) >> 2) << 2;
        const int nSrc = (ry1 - ry0) * nCols;
        // lane's source position, advanced incrementally from chunk to chunk (no per-chunk division)
        int sy, sx;
        {
            const int i0 = wv * CS + lane, r = i0 / nCols;           // wave wv starts with chunk wv
            sy = ry0 + r; sx = cx0 + (i0 - r * nCols);
        }
        const int xEnd = cx0 + nCols;
        // Per-visit operand addressing: entry u = min(|lane coordinate - source coordinate|, guard) of the lane's source table,
        // guard = the source's own batch radius + 1, where the table holds 0 (a lane whose row / column is out of the source's
        // reach reads it). With coordinates in bytes (x 4) the LDS address is min(|lane - source| + block, guard address) + 8
        // = v_sad_u32 (with the block address as its accumulator) + v_min_u32 per operand, the 8 as immediate offset of the read.
        // Coordinates carry a bias (64 rows, 128 columns) so that they are unsigned.
        const int ldsBase = (int)(size_t)(__attribute__((address_space(3))) float*)lds;   // LDS byte address of the slice
        int laneTab = ldsBase + kq * TS * 4;                         // + 16*q*TS: the lane's source block (source kq of the quad)
        const int laneRow4 = 4 * (oy0 + li - 32 - ry0 + 64);         // output row of the lane relative to the window's first source row (tile row t: source - 16 t)
        const int laneCol4 = 4 * (ox0 + li - kq - 32 - cx0 + 128);   // output column minus the lane's source offset in the quad (window-relative)
        // (opaque to the optimiser: otherwise it folds the per-visit scalar offset into these per-lane constants as
        //  (kq + q) * T and re-evaluates that with a quarter-rate v_mul_lo_u32 at every visit)
        asm volatile("" : "+v"(laneTab));
        // dose and 1/sigma of a chunk are fetched one chunk ahead (one memory round trip, hidden behind the previous chunk)
        const float* __restrict__ iddSlice = bevIdd + sliceOff;
        const float* __restrict__ rsSlice = bevRSigmaEff + sliceOff;
        float doseN = 0.0f, rsN = 0.0f;
        if (lane < CS && wv * CS + lane < nSrc && sx < cx1) {
            const unsigned int off = (unsigned int)(__mul24(sy, W) + sx) * 4u;         // byte offset within the slice: 32 bits suffice (W, H <= 4095: 24-bit multiply)
            doseN = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(iddSlice) + off);
            rsN = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rsSlice) + off);
        }
        int sxN = sx, syN = sy;
        for (int s0 = wv * CS; s0 < nSrc; s0 += kKsSplit * CS) {
            float dose = doseN;
            const float rs = rsN;
            sx = sxN; sy = syN;
            doseN = 0.0f; rsN = 0.0f;
            if (s0 + kKsSplit * CS < nSrc) {
                sxN += kKsSplit * CS; while (sxN >= xEnd) { sxN -= nCols; ++syN; }
                if (lane < CS && s0 + kKsSplit * CS + lane < nSrc && sxN < cx1) {
                    const unsigned int off = (unsigned int)(__mul24(syN, W) + sxN) * 4u;
                    doseN = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(iddSlice) + off);
                    rsN = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rsSlice) + off);
                }
            }
            if (!__any(dose != 0.0f)) continue;                      // chunk carries no dose: contributes exact zeros
            // ---- reach masks: which of the 8 output tiles can each source touch with ITS OWN batch radius ----
            int rhoS = -1, tmask = 0;
            if (dose != 0.0f) {
                rhoS = effT[__mul24((sy >> 3) - ty0, ntx) + ((sx >> 5) - tx0)];
                if (rhoS < 0) dose = 0.0f;
                else {
                    const int px = sx + 32 - ox0, py = sy + 32 - oy0;        // source position relative to the owned tile
                    const int tLo = max((px - rhoS) >> 4, 0), tHi = min((px + rhoS) >> 4, 3);
                    const int xm = tLo <= tHi ? (2 << tHi) - (1 << tLo) : 0; // bits tLo..tHi
                    const int rows = ((py + rhoS >= 0 && py - rhoS <= 15) ? 16 : 0) | ((py + rhoS >= 16 && py - rhoS <= 31) ? 32 : 0);
                    if (xm != 0 && rows != 0) tmask = xm | rows;
                }
            }
            // A quad's mask: bits 0..3 = tile columns some source of the quad reaches, bit 4 / 5 = upper / lower tile row (the
            // sources of a quad share row and batch radius, so the rows are common; a pair (row, column) that no source reaches
            // would only add the tables' zero entries): one MFMA per (row, column) pair, one single-bit scalar test each
            tmask |= __builtin_amdgcn_update_dpp(0, tmask, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            tmask |= __builtin_amdgcn_update_dpp(0, tmask, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
            // (the four lanes of a quad hold the same mask: one ballot bit per quad, bit 4*q)
            unsigned long long live = __ballot(tmask != 0) & 0x1111111111111111ull;
            // one readlane per visit: mask | biased source row x 4 (10 bits) | biased column in window x 4 (12 bits): two s_bfe decode them
            const int qinfo = tmask | ((sy - ry0 + 64) << 10) | ((sx - cx0 + 128) << 20);
            if (!live) continue;                                     // no source of the chunk reaches the tile
            // ---- build: weight table of one source per lane ----
            __builtin_amdgcn_wave_barrier();
            if (lane < CS) {
                // A source's table is read up to ITS OWN zero guard, entry rhoS + 1 (the lookups clamp to it), so the entries
                // beyond are never read and need no zeroing: the series below runs unmasked, a dead source (no dose / no
                // radius) only gets the guard at entry 0.
                const int guard = rhoS >= 0 ? rhoS + 1 : 0;
                float* sb = lds + lane * TS;
                sb[0] = dose;
                sb[1] = __int_as_float(ldsBase + (lane * TS + guard) * 4);   // LDS byte address of the guard entry, less the 8 bytes of this pair
                float* m = sb + 2;
                if (rhoS >= 0 && rs <= 0.5f) {
                    // Pixel-integrated Gaussian weights e_i = (1/2)(erf(rs(i+1/2)) - erf(rs(i-1/2))) (kernel_wrapper.cuh:459-467)
                    // evaluated as the Taylor series of the integral around the pixel centre x = rs*i:
                    //   e_i = rs/sqrt(pi) * exp(-x^2) * (1 + H2(x) rs^2/24 + H4(x) rs^4/1920 + H6(x) rs^6/322560),
                    //   H2 = 4x^2-2, H4 = 16x^4-48x^2+12, H6 = 64x^6-480x^4+720x^2-120  (g^(2n)/g of g = exp(-x^2)),
                    // collected into a cubic in w = i^2 with per-source coefficients (3 FMAs per entry), and exp(-x^2)
                    // advanced by the recurrence g_{i+1} = g_i q_i, q_{i+1} = q_i q_0^2, q_0 = exp(-rs^2).
                    // For rs <= 0.5 (sigma >= 1.4 ray pixels) the truncation is < 3e-8 absolute — the size of the rounding of
                    // the float erf DIFFERENCE itself (cancellation) — at ~10 vector instructions per entry instead of an erff
                    // with its exp (~45). Sharper sources (few entries) keep the erff form below.
                    const float h2 = rs * rs, h4 = h2 * h2;
                    const float k1 = h2 * (1.0f / 24.0f), k2 = h4 * (1.0f / 1920.0f), k3 = h4 * h2 * (1.0f / 322560.0f);
                    const float c0 = 1.0f - 2.0f * k1 + 12.0f * k2 - 120.0f * k3;
                    const float c1 = (4.0f * k1 - 48.0f * k2 + 720.0f * k3) * h2;
                    const float c2 = (16.0f * k2 - 480.0f * k3) * h4;
                    const float c3 = 64.0f * k3 * (h4 * h2);
                    // exp(-rs^2) on the hardware exp2 (h2 <= 0.25: no range reduction needed; <= 1 ulp like expf)
                    float q = __builtin_amdgcn_exp2f(-1.4426950409f * h2), gq = 0.5641895835f * rs;    // gq = rs/sqrt(pi) * exp(-x_i^2)
                    const float cq = q * q;
                    // two entries per trip (entry 0 first), so the LDS stores use immediate offsets
                    const float e0 = c0 * gq;
                    m[0] = e0;
                    gq *= q; q *= cq;
                    for (int i = 1; i <= Tm; i += 2) {
                        const float w0 = (float)(i * i), w1 = (float)((i + 1) * (i + 1));
                        const float s0 = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c3, w0, c2), w0, c1), w0, c0);
                        const float s1 = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c3, w1, c2), w1, c1), w1, c0);
                        const float g1 = gq * q, q1 = q * cq;
                        const float ea = gq * s0, eb = g1 * s1;
                        gq = g1 * q1; q = q1 * cq;
                        m[i] = ea;
                        if (i + 1 <= Tm) m[i + 1] = eb;
                    }
                } else if (rhoS >= 0) {
                    float erfNew = erff(rs * 0.5f), erfOld = -erfNew;
                    for (int i = 0; i <= rhoS; ++i) {
                        m[i] = 0.5f * (erfNew - erfOld);
                        erfOld = erfNew;
                        erfNew = erff(rs * ((float)i + 1.5f));
                    }
                }
                m[guard] = 0.0f;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- accumulate: one MFMA per (source quad, 16x16 output tile) pair whose bands intersect ----
            // Only quads that reach the tile are visited: their lanes are taken from the wave ballot (scalar bit scan), mask
            // and grid position from the lane that built the quad's first source (one readlane).
            // Three loops, one per row case of the quads (both tile rows, upper only, lower only): inside a loop the rows are
            // known at compile time, so a visit makes one scalar test per tile column and nothing else — the scalar / branch
            // path is what limits this loop, not the vector ALUs. (Sums are formed in this fixed order.)
            auto visits = [&](unsigned long long lv, auto rowsTag) {
                constexpr int ROWS = decltype(rowsTag)::value;       // bit 0: upper tile row, bit 1: lower tile row
                while (lv) {
                    const int q4 = __builtin_ctzll(lv);              // 4*q
                    asm("s_bitset0_b64 %0, %1" : "+s"(lv) : "s"(q4));   // lv &= ~(1 << q4)
                    const int qi = __builtin_amdgcn_readlane(qinfo, q4);   // column bits 0..3 are tested in place
                    int ctr;                                         // byte address of entry 0 of the lane's source table (one v_add per visit)
                    asm("v_add_u32 %0, %1, %2" : "=v"(ctr) : "s"(q4 * TS * 4), "v"(laneTab));
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 dg = *(__attribute__((address_space(3))) const f32x2*)(size_t)ctr;   // (dose, guard address) head the source block
                    const int ctrMax = __float_as_int(dg.y);         // the zero guard of that table
                    // scalar, biased, in bytes: bits 8..19 (8, 9 are zero) and bits 18..31 (18, 19 are zero: the row field stays below 256)
                    const int qRowB4 = (qi >> 8) & 0xFFF, qColB4 = (int)((unsigned)qi >> 18);
                    typedef __attribute__((address_space(3))) const float* lptr;
                    const float dl = dg.x;
                    auto entry = [&](int laneCoord4, int srcCoord4) -> float {
                        unsigned int u;
                        asm("v_sad_u32 %0, %1, %2, %3" : "=v"(u) : "v"(laneCoord4), "s"(srcCoord4), "v"(ctr));
                        u = u < (unsigned)ctrMax ? u : (unsigned)ctrMax;
                        return *(lptr)(size_t)(u + 8);                // entries follow the pair (immediate offset of the LDS read)
                    };
                    float a0 = 0.0f, a1 = 0.0f;                      // A = dose * m[|row - y_s|]
                    if (ROWS & 1) a0 = dl * entry(laneRow4, qRowB4);
                    if (ROWS & 2) a1 = dl * entry(laneRow4, qRowB4 - 64);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (qi & (1 << t)) {
                            const float bt = entry(laneCol4, qColB4 - 64 * t);
                            if (ROWS & 1) acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bt, acc[0][t], 0, 0, 0);
                            if (ROWS & 2) acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bt, acc[1][t], 0, 0, 0);
                        }
                    }
                }
            };
            const int rowBits = tmask & 48;
            visits(live & __ballot(rowBits == 48), std::integral_constant<int, 3>{});
            visits(live & __ballot(rowBits == 16), std::integral_constant<int, 1>{});
            visits(live & __ballot(rowBits == 32), std::integral_constant<int, 2>{});
        }
    }
    // ---- the item's waves add their accumulators in fixed order (wave 0 + wave 1 + ...) through LDS ----
    if (kKsSplit > 1) {
        __syncthreads();                                             // every wave is done with its tables
        for (int r = 1; r < kKsSplit; ++r) {
            if (wv == r) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int c = 0; c < 4; ++c) ldsAll[((a * 4 + b) * 4 + c) * 64 + lane] = acc[a][b][c];
            }
            __syncthreads();
            if (wv == 0) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[a][b][c] += ldsAll[((a * 4 + b) * 4 + c) * 64 + lane];
            }
            if (r + 1 < kKsSplit) __syncthreads();
        }
        if (wv != 0) return;
    }
    // ---- epilogue: the BEV dose of (tile, slice k) is the sum over the layer groups that deposit at k. ----
    // No second pass over the partials: the groups' accumulators are added in a fixed binary tree over the ranks of the active
    // groups. At every node the LATER of the two arriving waves does the addition (a + b = b + a exactly, so the result does not
    // depend on who that is: bitwise reproducible), the earlier one has left its data in a slot and exits. Hand-off between
    // waves on different XCDs (L2s are not coherent with each other): the data stores and loads are agent-scope (sc1: at the
    // memory side), drained with s_waitcnt vmcnt(0) before the node counter (an atomic at the memory side) is touched —
    // MI355X_MICROARCH.md, "Correctness boundaries". A wave never waits for another: every item runs to its end on its own.
    int nAct = 0, rank = 0;
    for (int g2 = 0; g2 < G; ++g2) { const int a = k < st->groupPassive[g2] ? 1 : 0; nAct += a; rank += (g2 < g) ? a : 0; }
    constexpr int kSlotFloats = kKsTileX * kKsTileY;                 // 2048: [element 0..31][lane]
    const size_t nT = (size_t)nTX * nTY;
    int* cnt = nodeCount + ((size_t)tile * fc.S + k) * 32;
    int level = 0, n = nAct;
    while (n > 1) {
        const int sib = rank ^ 1;
        if (sib < n) {
            float* mine = bevPart + (((size_t)(rank << level) * fc.S + k) * nT + tile) * kSlotFloats;
#pragma unroll
            for (int e = 0; e < 32; ++e) __hip_atomic_store(mine + e * 64 + lane, acc[e >> 4][(e >> 2) & 3][e & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the slot is complete at the memory side
            int* node = cnt + (32 - (32 >> level)) + (rank >> 1);
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(node, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == 0) return;                                    // first at this node: the sibling takes over
            if (lane == 0) __hip_atomic_store(node, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            const float* theirs = bevPart + (((size_t)(sib << level) * fc.S + k) * nT + tile) * kSlotFloats;
#pragma unroll
            for (int h = 0; h < 4; ++h) {                            // (in quarters: 8 loads in flight, within the kernel's 72 registers)
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = __hip_atomic_load(theirs + (8 * h + e) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[h >> 1][2 * (h & 1) + (e >> 2)][e & 3] += t[e];
            }
        }
        rank >>= 1; n = (n + 1) >> 1; ++level;
    }
    // root: one plain store per element; D[row=(lane>>4)*4+reg][col=lane&15]
    float* out = bevDose + (size_t)k * fc.bevW * fc.bevH;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
        for (int tx = 0; tx < 4; ++tx)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int oy = oy0 + 16 * ty + 4 * kq + reg, ox = ox0 + 16 * tx + li;
                if (oy < fc.bevH && ox < fc.bevW) out[(size_t)oy * fc.bevW + ox] = acc[ty][tx][reg];
            }
}

// ------------------------------------------------------------------------------------------------
// K7u, the superposition of a field whose every (layer, step) slice has ONE sigma over its live rays (a water phantom, the reference's
// own WATER_CUBE_TEST): rtd_uniform.hpp. Whether a field qualifies is decided on the device (k_fill / k_ks_plan:
// FieldState::uniformField); that launch returns at once otherwise, and the general superposition does when it does.
constexpr int kUniMaxBevH = 256;                                     // a slice has at most 16 row blocks of 16 rows

// ------------------------------------------------------------------------------------------------
// K8: fan -> dose-grid transfer = primTransfDiv (kernel_wrapper.cu:69-97). The reference copies the BEV slab
// into a 3-D texture first (:1107-1141); here the trilinear BORDER sample is taken from the BEV buffer itself
// (slab origin and extent applied in index arithmetic), which removes that copy. One thread per dose (x,y)
// column and z-chunk inside the device-side bounding box (getFanIdx(z) is closed-form, so z splits freely).
struct ClipBox { int lo[3], hi[3]; };          // inclusive dose-index box a transfer / clear is restricted to (a GPU's slab of the volume)

// INIT: the voxels of the field's dose box are WRITTEN (dose or zero) instead of accumulated into: the first field of a plan
// then needs neither a cleared box nor the read half of the read-modify-write (rtd_field_transfer_init).
template <bool INIT>
__global__ __launch_bounds__(256) void k_transfer(float* __restrict__ dose, int nx, int ny, int nz,
                                                   const float* __restrict__ bevDose, const FieldState* __restrict__ st,
                                                   FieldConst fc, int zChunk, ClipBox clip) {
    const int first = st->beamFirstInside;
    const int slabZ = st->firstCalculatedPassive - first;
    if (slabZ <= 0 || st->errorFlags) return;                        // on a device-side error the dose volume stays untouched
    // The box that can receive dose is known on the device only: a fixed grid of blocks strides over its 32 x 8 x zChunk
    // bricks (a grid over the whole dose volume would be mostly blocks that load the box and exit — measured 55 of 137 us).
    const int bx0 = max(st->tboxMin[0], clip.lo[0]), by0 = max(st->tboxMin[1], clip.lo[1]), bz0 = max(st->tboxMin[2], clip.lo[2]);
    const int bx1 = min(st->tboxMax[0], clip.hi[0]), by1 = min(st->tboxMax[1], clip.hi[1]), bz1 = min(st->tboxMax[2], clip.hi[2]);
    if (bx1 < bx0 || by1 < by0 || bz1 < bz0) return;
    const int nbx = (bx1 - bx0) / 32 + 1, nby = (by1 - by0) / 8 + 1, nbz = (bz1 - bz0) / zChunk + 1;
    const int nBricks = nbx * nby * nbz;
    const TransferParams p0 = st->transfer;
    const int pW = st->packW, pH = st->packH;
    const float pX0 = (float)st->packX0, pY0 = (float)st->packY0;    // (subtracting an integer below the coordinate is exact)
    const float* slab = bevDose + (size_t)st->slabFirst * pW * pH;
    // outside this rectangle (+1 for the interpolation neighbours) every BEV slice is exactly zero: no loads needed
    const float exLo = (float)(st->bevLo[0] - 1), exHi = (float)(st->bevHi[0] + 1), eyLo = (float)(st->bevLo[1] - 1), eyHi = (float)(st->bevHi[1] + 1);
    const size_t nxy = (size_t)nx * ny;
    for (int brick = blockIdx.x; brick < nBricks; brick += gridDim.x) {
        const int bx = brick % nbx, by = (brick / nbx) % nby, bz = brick / (nbx * nby);
        const int x = bx0 + 32 * bx + threadIdx.x, y = by0 + 8 * by + threadIdx.y;
        const int z0 = bz0 + bz * zChunk, z1 = min(z0 + zChunk - 1, bz1);
        // (culling whole bricks in the empty corners of an oblique beam's box with an 8-corner test measured slower at every
        //  angle — 0.095 vs 0.084 ms at 0 degrees, 0.132 vs 0.128 at 45: those bricks already cost one position per voxel only.
        //  Also measured slower, parity-green: the 32 loads of the four samples issued before the first use (0.094 ms, 96 VGPRs),
        //  and the brick's BEV cells staged in LDS so that the gathers hit LDS (0.104 ms; 768^3: 0.257 vs 0.209) — the kernel is
        //  not bound by the gathers.)
        if (x > bx1 || y > by1) continue;                            // (the box lies inside the dose grid)
        TransferParams p = p0;
        p.init(x, y);
        float* res = dose + (size_t)z0 * nxy + (size_t)y * nx + x;
        // four depth samples per trip: their 32 BEV loads are in flight together before the dose read-modify-writes
        constexpr int kZU = 4;
        for (int z = z0; z <= z1; z += kZU) {
            float tmp[kZU];
#pragma unroll
            for (int u = 0; u < kZU; ++u) {
                tmp[u] = 0.0f;
                if (z + u <= z1) {
                    Vec3 pos = p.getFanIdx(z + u);
                    if (pos.x > exLo && pos.x < exHi && pos.y > eyLo && pos.y < eyHi)
                        tmp[u] = sample3dBorder(slab, pW, pH, slabZ, pos.x - pX0, pos.y - pY0, pos.z);
                }
            }
#pragma unroll
            for (int u = 0; u < kZU; ++u) {
                if (INIT) { if (z + u <= z1) res[u * nxy] = tmp[u] > 0.0f ? tmp[u] : 0.0f; }
                else if (tmp[u] > 0.0f) res[u * nxy] += tmp[u];
            }
            res += kZU * nxy;
        }
    }
}

// The same transfer for beams that run along the dose x axis (gantry near 90 / 270 degrees): there x-adjacent voxels lie in
// different BEV slices and the gathers of k_transfer touch 64 slices per load (measured 0.21 ms against 0.084 ms at 0 degrees).
// Here the lanes of the gather phase run along the dose axis B (1 = y, 2 = z) that maps to BEV x; the values cross an LDS
// tile and are added to the dose with lanes along x again. Per voxel the arithmetic is that of k_transfer.
// (Gathering 8 x 8 patches of the (x, B) plane per wave for oblique beams measured within 3 % of this kernel at 45 degrees.)
template <int B, bool INIT>
__global__ __launch_bounds__(256) void k_transfer_t(float* __restrict__ dose, int nx, int ny, int nz,
                                                     const float* __restrict__ bevDose, const FieldState* __restrict__ st,
                                                     FieldConst fc, int cChunk, ClipBox clip) {
    constexpr int C = B == 1 ? 2 : 1;                                // the axis a thread walks
    constexpr int kZU = 4;
    __shared__ float tile[2][kZU][16][17];
    const int first = st->beamFirstInside;
    const int slabZ = st->firstCalculatedPassive - first;
    if (slabZ <= 0 || st->errorFlags) return;                        // on a device-side error the dose volume stays untouched
    const int lo[3] = {max(st->tboxMin[0], clip.lo[0]), max(st->tboxMin[1], clip.lo[1]), max(st->tboxMin[2], clip.lo[2])};
    const int hi[3] = {min(st->tboxMax[0], clip.hi[0]), min(st->tboxMax[1], clip.hi[1]), min(st->tboxMax[2], clip.hi[2])};
    if (hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]) return;
    const int nbx = (hi[0] - lo[0]) / 16 + 1, nbb = (hi[B] - lo[B]) / 16 + 1, nbc = (hi[C] - lo[C]) / cChunk + 1;
    const int nBricks = nbx * nbb * nbc;
    const TransferParams p0 = st->transfer;
    const int pW = st->packW, pH = st->packH;
    const float pX0 = (float)st->packX0, pY0 = (float)st->packY0;
    const float* slab = bevDose + (size_t)st->slabFirst * pW * pH;
    const float exLo = (float)(st->bevLo[0] - 1), exHi = (float)(st->bevHi[0] + 1), eyLo = (float)(st->bevLo[1] - 1), eyHi = (float)(st->bevHi[1] + 1);
    const size_t nxy = (size_t)nx * ny;
    const size_t strideB = B == 1 ? (size_t)nx : nxy, strideC = C == 1 ? (size_t)nx : nxy;
    const int tid = threadIdx.y * 32 + threadIdx.x;
    const int gB = tid & 15, gX = tid >> 4;                          // gather phase: lanes along B
    const int aX = tid & 15, aB = tid >> 4;                          // add phase: lanes along x
    int buf = 0;
    for (int brick = blockIdx.x; brick < nBricks; brick += gridDim.x) {
        const int bx = brick % nbx, bb = (brick / nbx) % nbb, bc = brick / (nbx * nbb);
        const int x0 = lo[0] + 16 * bx, b0 = lo[B] + 16 * bb;
        const int c0 = lo[C] + bc * cChunk, c1 = min(c0 + cChunk - 1, hi[C]);
        const int xg = x0 + gX, bg = b0 + gB;
        const bool gIn = xg <= hi[0] && bg <= hi[B];
        const int xa = x0 + aX, ba = b0 + aB;
        const bool aIn = xa <= hi[0] && ba <= hi[B];
        TransferParams p = p0;
        if (B == 2) p.init(xg, 0);                                   // y is walked: start is rebuilt per sample below
        else p.init(xg, bg);
        float* res = dose + (size_t)c0 * strideC + (size_t)ba * strideB + xa;
        for (int c = c0; c <= c1; c += kZU) {
#pragma unroll
            for (int u = 0; u < kZU; ++u) {
                float v = 0.0f;
                if (gIn && c + u <= c1) {
                    Vec3 pos;
                    if (B == 2) { TransferParams q = p0; q.init(xg, c + u); pos = q.getFanIdx(bg); }
                    else pos = p.getFanIdx(c + u);
                    if (pos.x > exLo && pos.x < exHi && pos.y > eyLo && pos.y < eyHi)
                        v = sample3dBorder(slab, pW, pH, slabZ, pos.x - pX0, pos.y - pY0, pos.z);
                }
                tile[buf][u][gB][gX] = v;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kZU; ++u) {
                const float v = tile[buf][u][aB][aX];
                if (INIT) { if (aIn && c + u <= c1) res[u * strideC] = v > 0.0f ? v : 0.0f; }
                else if (aIn && v > 0.0f) res[u * strideC] += v;
            }
            res += kZU * strideC;
            buf ^= 1;                                                // the other tile is free: its readers passed the barrier above
        }
    }
}

// Several fields into one box in ONE pass (rtd_fields_transfer_init): every voxel of `box` is WRITTEN with
// ((0 + field 0) + field 1) + ... — the positive samples in list order, exactly the values that rtd_field_transfer of each field in
// turn would have accumulated into a zeroed volume (same samples, same order of the float additions), without the N - 1
// read-modify-write passes over the volume, without a clear, in one launch. What a GPU of a multi-GPU plan does with the BEV slabs
// it gathered (its slab of the volume = box), and why it exists: N clipped launches per plan step measured 2x the per-voxel
// cost of one full launch.
// Block = one 16^3 brick; thread (x, y) of the brick keeps its 16 z sums in a private LDS column (registers would have to be
// indexed dynamically by the rolled chunk loops). A field is sampled with the lanes along the dose axis that moves fastest along its
// BEV x (its transferMode, as k_transfer / k_transfer_t): mode 0 directly, modes 1 / 2 through an LDS tile that turns the
// gather layout into the (x, y) layout of the sums.
constexpr int kMultiMaxFields = 16;
struct MultiFields {
    const float* bev[kMultiMaxFields];
    const FieldState* st[kMultiMaxFields];
    int mode[kMultiMaxFields];
    int n;
};

// What a brick needs to know of a field, gathered once per block (thread f reads field f's state record: one memory round trip for
// all fields instead of one per (brick, field) — with ~1 brick per block and 8 fields that latency was comparable to the sampling).
struct MultiParam {
    int valid, slabZ;
    int box0[3], box1[3];
    TransferParams tp;
    int pW, pH;
    float pX0, pY0, exLo, exHi, eyLo, eyHi;
    unsigned int slabOff;                                             // floats from the slab pointer to its first slice
};
static_assert(sizeof(MultiParam) % 4 == 0, "MultiParam is copied word by word");

__global__ __launch_bounds__(256, 6) void k_transfer_multi(float* __restrict__ dose, int nx, int ny, int nz, MultiFields mf, ClipBox box) {
    __shared__ float accT[16][256];                                   // [z][thread]: a thread's 16 sums (private column: no barriers needed)
    __shared__ float tile[4][16][17];
    __shared__ MultiParam sPar[kMultiMaxFields];
    if ((int)threadIdx.x < mf.n) {
        const FieldState* st = mf.st[threadIdx.x];
        MultiParam q;
        const int first = st->beamFirstInside;
        q.slabZ = st->firstCalculatedPassive - first;
        q.valid = (q.slabZ > 0 && !st->errorFlags) ? 1 : 0;           // as k_transfer: otherwise the field deposits nothing
        // the field's own dose box, cut to the written box: the voxels of a partial brick beyond it are neither sampled nor written
        for (int a = 0; a < 3; ++a) { q.box0[a] = st->tboxMin[a]; q.box1[a] = min(st->tboxMax[a], box.hi[a]); }
        q.tp = st->transfer;
        q.pW = st->packW; q.pH = st->packH; q.pX0 = (float)st->packX0; q.pY0 = (float)st->packY0;
        q.exLo = (float)(st->bevLo[0] - 1); q.exHi = (float)(st->bevHi[0] + 1); q.eyLo = (float)(st->bevLo[1] - 1); q.eyHi = (float)(st->bevHi[1] + 1);
        q.slabOff = (unsigned int)st->slabFirst * (unsigned int)q.pW * (unsigned int)q.pH;
        sPar[threadIdx.x] = q;
    }
    __syncthreads();
    const int nbx = (box.hi[0] - box.lo[0]) / 16 + 1, nby = (box.hi[1] - box.lo[1]) / 16 + 1, nbz = (box.hi[2] - box.lo[2]) / 16 + 1;
    const int nBricks = nbx * nby * nbz;
    const int tid = threadIdx.x;
    const int aX = tid & 15, aY = tid >> 4;                           // layout of the sums: lanes along x
    const int gB = tid & 15, gX = tid >> 4;                           // layout of the gathers of modes 1 / 2: lanes along B
    const size_t nxy = (size_t)nx * ny;
    for (int brick = blockIdx.x; brick < nBricks; brick += gridDim.x) {
        const int x0 = box.lo[0] + 16 * (brick % nbx), y0 = box.lo[1] + 16 * ((brick / nbx) % nby), z0 = box.lo[2] + 16 * (brick / (nbx * nby));
#pragma unroll
        for (int z = 0; z < 16; ++z) accT[z][tid] = 0.0f;
        for (int fi = 0; fi < mf.n; ++fi) {
            // the field's record from LDS into scalar registers (the values are block-uniform)
            MultiParam q;
            {
                const int* src = reinterpret_cast<const int*>(&sPar[fi]);
                int* dst = reinterpret_cast<int*>(&q);
#pragma unroll
                for (int w = 0; w < (int)(sizeof(MultiParam) / 4); ++w) dst[w] = __builtin_amdgcn_readfirstlane(src[w]);
            }
            if (!q.valid) continue;                                   // (uniform)
            const int slabZ = q.slabZ;
            const int bx0 = q.box0[0], by0 = q.box0[1], bz0 = q.box0[2], bx1 = q.box1[0], by1 = q.box1[1], bz1 = q.box1[2];
            if (x0 > bx1 || x0 + 15 < bx0 || y0 > by1 || y0 + 15 < by0 || z0 > bz1 || z0 + 15 < bz0) continue;   // (uniform)
            const TransferParams p0 = q.tp;
            const int pW = q.pW, pH = q.pH;
            const float pX0 = q.pX0, pY0 = q.pY0;
            const float* slab = mf.bev[fi] + q.slabOff;
            const float exLo = q.exLo, exHi = q.exHi, eyLo = q.eyLo, eyHi = q.eyHi;
            auto sampleAt = [&](const Vec3& pos) -> float {
                if (pos.x > exLo && pos.x < exHi && pos.y > eyLo && pos.y < eyHi)
                    return sample3dBorder(slab, pW, pH, slabZ, pos.x - pX0, pos.y - pY0, pos.z);
                return 0.0f;
            };
            const int mode = mf.mode[fi];
            if (mode == 0) {
                const int x = x0 + aX, y = y0 + aY;
                const bool in = x >= bx0 && x <= bx1 && y >= by0 && y <= by1;
                TransferParams p = p0;
                p.init(x, y);
#pragma unroll 1
                for (int c = 0; c < 16; c += 4) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int z = z0 + c + u;
                        v[u] = (in && z >= bz0 && z <= bz1) ? sampleAt(p.getFanIdx(z)) : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (v[u] > 0.0f) accT[c + u][tid] += v[u];
                }
            } else if (mode == 1) {
                // lanes along y, z walked: tile[u][y][x]
                const int x = x0 + gX, y = y0 + gB;
                const bool in = x >= bx0 && x <= bx1 && y >= by0 && y <= by1;
                TransferParams p = p0;
                p.init(x, y);
#pragma unroll 1
                for (int c = 0; c < 16; c += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int z = z0 + c + u;
                        tile[u][gB][gX] = (in && z >= bz0 && z <= bz1) ? sampleAt(p.getFanIdx(z)) : 0.0f;
                    }
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const float v = tile[u][aY][aX]; if (v > 0.0f) accT[c + u][tid] += v; }
                    __syncthreads();
                }
            } else {
                // lanes along z, y walked: tile[u][z][x]; the wave that owns rows c .. c+3 of the brick collects a chunk
                const int x = x0 + gX, z = z0 + gB;
                const bool in = x >= bx0 && x <= bx1 && z >= bz0 && z <= bz1;
#pragma unroll 1
                for (int c = 0; c < 16; c += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int y = y0 + c + u;
                        float v = 0.0f;
                        if (in && y >= by0 && y <= by1) { TransferParams q = p0; q.init(x, y); v = sampleAt(q.getFanIdx(z)); }
                        tile[u][gB][gX] = v;
                    }
                    __syncthreads();
                    if ((aY >> 2) == (c >> 2)) {
#pragma unroll
                        for (int zz = 0; zz < 16; ++zz) { const float v = tile[aY & 3][zz][aX]; if (v > 0.0f) accT[zz][tid] += v; }
                    }
                    __syncthreads();
                }
            }
        }
        const int x = x0 + aX, y = y0 + aY;
        if (x <= box.hi[0] && y <= box.hi[1]) {
            float* res = dose + (size_t)z0 * nxy + (size_t)y * nx + x;
#pragma unroll
            for (int z = 0; z < 16; ++z) if (z0 + z <= box.hi[2]) res[z * nxy] = accT[z][tid];
        }
    }
}

// Zeroes the bricks of the dose box of the last transfer (rtd_field_clear_dose): same brick walk as k_transfer.
__global__ __launch_bounds__(256) void k_clear_box(float* __restrict__ dose, int nx, int ny, const FieldState* __restrict__ st, int zChunk,
                                                    ClipBox clip) {
    const int bx0 = max(st->tboxMin[0], clip.lo[0]), by0 = max(st->tboxMin[1], clip.lo[1]), bz0 = max(st->tboxMin[2], clip.lo[2]);
    const int bx1 = min(st->tboxMax[0], clip.hi[0]), by1 = min(st->tboxMax[1], clip.hi[1]), bz1 = min(st->tboxMax[2], clip.hi[2]);
    if (bx1 < bx0 || by1 < by0 || bz1 < bz0) return;
    const int nbx = (bx1 - bx0) / 32 + 1, nby = (by1 - by0) / 8 + 1, nbz = (bz1 - bz0) / zChunk + 1;
    const size_t nxy = (size_t)nx * ny;
    for (int brick = blockIdx.x; brick < nbx * nby * nbz; brick += gridDim.x) {
        const int bx = brick % nbx, by = (brick / nbx) % nby, bz = brick / (nbx * nby);
        const int x = bx0 + 32 * bx + threadIdx.x, y = by0 + 8 * by + threadIdx.y;
        const int z0 = bz0 + bz * zChunk, z1 = min(z0 + zChunk - 1, bz1);
        if (x > bx1 || y > by1) continue;
        float* res = dose + (size_t)z0 * nxy + (size_t)y * nx + x;
        for (int z = z0; z <= z1; ++z, res += nxy) *res = 0.0f;
    }
}

// Packs what another GPU needs to finish this field — the state record and the block of the BEV dose that can be non-zero
// (rectangle [bevLo-1, bevHi+1] of the slices [entry, passive)) — into one message: [FieldState, padded to kPackHeader
// bytes][slices x rows x columns]. The receiver runs k_transfer / k_transfer_t straight on the message (the header IS its
// state record, with the slab geometry rewritten), restricted to its own slab of the dose volume. The BEV block of a 512^3 field
// is ~10 MB against 60-83 MB for the dose box it turns into: the exchange of a multi-GPU plan is done in beam's-eye view.
constexpr int kPackHeader = 4096;
static_assert(sizeof(FieldState) <= kPackHeader, "the state record must fit the message header");
__global__ __launch_bounds__(256) void k_pack_bev(const float* __restrict__ bevDose, const FieldState* __restrict__ st, FieldConst fc,
                                                   unsigned char* __restrict__ msg, size_t capacity) {
    const int first = st->beamFirstInside, nz = max(st->firstCalculatedPassive - first, 0);
    const int x0 = max(st->bevLo[0] - 1, 0) & ~3, x1 = min(st->bevHi[0] + 1, fc.bevW - 1);     // columns in whole float4 (bevW % 32 == 0)
    const int y0 = max(st->bevLo[1] - 1, 0), y1 = min(st->bevHi[1] + 1, fc.bevH - 1);
    const bool none = nz == 0 || x1 < x0 || y1 < y0;
    const int w4 = none ? 0 : (x1 - x0 + 4) / 4, h = none ? 0 : y1 - y0 + 1;
    const size_t need = (size_t)kPackHeader + (size_t)nz * h * w4 * 16;
    const bool fits = need <= capacity;
    FieldState* hd = reinterpret_cast<FieldState*>(msg);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        FieldState s = *st;
        s.packX0 = x0; s.packY0 = y0; s.packW = 4 * w4; s.packH = h; s.slabFirst = 0;
        if (none) { s.tboxMin[0] = 0; s.tboxMax[0] = -1; }
        if (!fits) s.errorFlags |= kErrPackOverflow;                 // the receiver's transfer then leaves the dose untouched
        *hd = s;
    }
    if (!fits || none) return;
    float4* dst = reinterpret_cast<float4*>(msg + kPackHeader);
    const size_t P = (size_t)fc.bevW * fc.bevH, n4 = (size_t)nz * h * w4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % w4), r = (int)((i / w4) % h), k = (int)(i / ((size_t)w4 * h));
        dst[i] = *reinterpret_cast<const float4*>(bevDose + (size_t)(first + k) * P + (size_t)(y0 + r) * fc.bevW + x0 + 4 * c);
    }
}

}  // namespace rtd
