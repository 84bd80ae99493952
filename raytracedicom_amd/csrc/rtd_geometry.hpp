// rtd_geometry.hpp — value types of the dose engine shared by host and device code.
//
// MI355X-native re-statement of the reference's boundary value types and per-kernel parameter PODs:
//   Matrix3x3                    src/matrix_3x3.cu
//   Float3AffineTransform        src/float3_affine_transform.cu
//   Float3IdxTransform           src/float3_idx_transform.cu
//   Float3FromFanTransform       src/float3_from_fan_transform.cu
//   Float3ToFanTransform         src/float3_to_fan_transform.cu
//   DensityAndSpTracerParams     src/density_and_sp_tracer_params.cu
//   FillIddAndSigmaParams        src/fill_idd_and_sigma_params.cu
//   TransferParamStructDiv3      src/transfer_param_struct_div3.cu
// Plain structs, no CUDA/HIP vector headers. Operand order follows the reference expressions so that the
// results equal a float-exact evaluation of the reference formulas (this TU is built with -ffp-contract=off).
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define RTD_HD __host__ __device__ inline
#else
#define RTD_HD inline
#endif

namespace rtd {

struct Vec2 { float x, y; };
struct Vec3 { float x, y, z; };

RTD_HD Vec3 v3(float x, float y, float z) { Vec3 r; r.x = x; r.y = y; r.z = z; return r; }
RTD_HD Vec3 operator+(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTD_HD Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RTD_HD Vec3 operator*(Vec3 a, Vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RTD_HD Vec3 operator/(Vec3 a, Vec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
RTD_HD Vec3 operator*(Vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RTD_HD Vec3 operator*(float s, Vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
RTD_HD Vec3 operator/(Vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RTD_HD float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct Mat3 {  // row-major rows (matrix_3x3.cuh)
    Vec3 r0, r1, r2;
};
RTD_HD Mat3 transpose(Mat3 a) {  // matrix_3x3.cu:55-57
    Mat3 r;
    r.r0 = v3(a.r0.x, a.r1.x, a.r2.x); r.r1 = v3(a.r0.y, a.r1.y, a.r2.y); r.r2 = v3(a.r0.z, a.r1.z, a.r2.z);
    return r;
}
RTD_HD Vec3 operator*(Mat3 m, Vec3 a) { return v3(dot(m.r0, a), dot(m.r1, a), dot(m.r2, a)); }  // :23-26
RTD_HD Mat3 operator*(Mat3 m1, Mat3 m) {  // :28-34
    Mat3 m2 = transpose(m), r;
    r.r0 = v3(dot(m1.r0, m2.r0), dot(m1.r0, m2.r1), dot(m1.r0, m2.r2));
    r.r1 = v3(dot(m1.r1, m2.r0), dot(m1.r1, m2.r1), dot(m1.r1, m2.r2));
    r.r2 = v3(dot(m1.r2, m2.r0), dot(m1.r2, m2.r1), dot(m1.r2, m2.r2));
    return r;
}
RTD_HD float det(Mat3 a) {  // :42-44
    return (a.r0.x * (a.r1.y * a.r2.z - a.r1.z * a.r2.y) - a.r0.y * (a.r1.x * a.r2.z - a.r1.z * a.r2.x)
            + a.r0.z * (a.r1.x * a.r2.y - a.r1.y * a.r2.x));
}
inline Mat3 inverse(Mat3 a) {  // :46-53 (host only: 1./det is a double division in the reference)
    float oneOverDet = (float)(1.0 / (double)det(a));
    Mat3 adj, d;
    adj.r0 = v3(a.r1.y * a.r2.z - a.r1.z * a.r2.y, a.r0.z * a.r2.y - a.r0.y * a.r2.z, a.r0.y * a.r1.z - a.r0.z * a.r1.y);
    adj.r1 = v3(a.r1.z * a.r2.x - a.r1.x * a.r2.z, a.r0.x * a.r2.z - a.r0.z * a.r2.x, a.r0.z * a.r1.x - a.r0.x * a.r1.z);
    adj.r2 = v3(a.r1.x * a.r2.y - a.r1.y * a.r2.x, a.r0.y * a.r2.x - a.r0.x * a.r2.y, a.r0.x * a.r1.y - a.r0.y * a.r1.x);
    d.r0 = v3(oneOverDet, 0.0f, 0.0f); d.r1 = v3(0.0f, oneOverDet, 0.0f); d.r2 = v3(0.0f, 0.0f, oneOverDet);
    return adj * d;  // scalar -> diagonal Matrix3x3 (matrix_3x3.cu:17-19), then matrix product
}

struct Affine { Mat3 m; Vec3 v; };     // Float3AffineTransform
struct IdxTransform { Vec3 delta, offset; };  // Float3IdxTransform

RTD_HD Vec3 transformPoint(const Affine& a, Vec3 p) { return a.m * p + a.v; }              // float3_affine_transform.cu:14
inline Affine inverse(const Affine& a) { Affine r; r.m = inverse(a.m); r.v = inverse(a.m) * (a.v * -1.0f); return r; }  // :20-23
inline Affine concat(const Affine& t1, const Affine& t2) { Affine r; r.m = t2.m * t1.m; r.v = t2.m * t1.v + t2.v; return r; }  // :42-45
RTD_HD Vec3 transformPoint(const IdxTransform& t, Vec3 p) { return p * t.delta + t.offset; }  // float3_idx_transform.cu:17
inline IdxTransform inverse(const IdxTransform& t) {                                           // :19
    IdxTransform r; r.delta = v3(1.0f, 1.0f, 1.0f) / t.delta; r.offset = (-1.0f * t.offset) / t.delta; return r;
}
inline IdxTransform shiftOffset(const IdxTransform& t, Vec3 s) { IdxTransform r = t; r.offset = t.offset + s; return r; }  // :21

struct FromFan { IdxTransform fitf; Affine gtii; Vec2 dist; };   // Float3FromFanTransform
struct ToFan { Affine iitg; Vec2 dist; IdxTransform ftfi; };     // Float3ToFanTransform

RTD_HD Vec3 transformPoint(const FromFan& t, Vec3 fanIdx) {  // float3_from_fan_transform.cu:37-42
    Vec3 interm = transformPoint(t.fitf, fanIdx);
    interm.x *= 1.0f - interm.z / t.dist.x;
    interm.y *= 1.0f - interm.z / t.dist.y;
    return transformPoint(t.gtii, interm);
}
RTD_HD Vec3 transformPoint(const ToFan& t, Vec3 imIdx) {  // float3_to_fan_transform.cu:26-31
    Vec3 interm = transformPoint(t.iitg, imIdx);
    interm.x /= 1.0f - interm.z / t.dist.x;
    interm.y /= 1.0f - interm.z / t.dist.y;
    return transformPoint(t.ftfi, interm);
}
inline ToFan invertAndShift(const FromFan& t, Vec3 shift) {  // float3_from_fan_transform.cu:27-30
    ToFan r; r.iitg = inverse(t.gtii); r.dist = t.dist; r.ftfi = shiftOffset(inverse(t.fitf), shift); return r;
}

// ---- DensityAndSpTracerParams (density_and_sp_tracer_params.cu:15-36) ----
struct TracerParams {
    float densityScale, spScale;
    unsigned int steps;
    Vec3 coefOffset, coefIdxI, coefIdxJ, transl, corner, delta;
    Vec2 dist;

    RTD_HD Vec3 getStart(int i, int j) const {
        return (float(i) * coefIdxI) * (1.0f - corner.z / dist.x) + (float(j) * coefIdxJ) * (1.0f - corner.z / dist.y) + transl;
    }
    RTD_HD Vec3 getInc(int i, int j) const {
        return (coefOffset - (float(i) * coefIdxI) / dist.x - (float(j) * coefIdxJ) / dist.y) * delta.z;
    }
    RTD_HD float stepLen(int i, int j) const {
        float deltaX = (corner.x + float(i) * delta.x) / dist.x;
        float deltaY = (corner.y + float(j) * delta.y) / dist.y;
        return fabsf(delta.z) * sqrtf(1.0f + deltaX * deltaX + deltaY * deltaY);
    }
};
inline TracerParams makeTracerParams(float densityScaleFact, float spScaleFact, unsigned int steps, const FromFan& t) {
    TracerParams p;
    p.densityScale = densityScaleFact; p.spScale = spScaleFact; p.steps = steps;
    p.dist = t.dist; p.corner = t.fitf.offset; p.delta = t.fitf.delta;
    Mat3 tT = transpose(t.gtii.m);
    p.coefOffset = tT.r2 - (tT.r0 * p.corner.x) / p.dist.x - (tT.r1 * p.corner.y) / p.dist.y;
    p.coefIdxI = tT.r0 * p.delta.x;
    p.coefIdxJ = tT.r1 * p.delta.y;
    p.transl = t.gtii.v + tT.r2 * p.corner.z + (tT.r0 * p.corner.x) * (1.0f - p.corner.z / p.dist.x)
               + (tT.r1 * p.corner.y) * (1.0f - p.corner.z / p.dist.y);
    return p;
}

// ---- FillIddAndSigmaParams, geometry part shared by all layers (fill_idd_and_sigma_params.cu:12-72) ----
struct FillGeom {
    Vec3 corner, delta;
    Vec2 dist;
    float volConst, volLin, volSq, stepLength, rRlScale;

    RTD_HD Vec2 voxelWidth(unsigned int k) const {  // :42-46
        Vec2 r;
        r.x = delta.x * (1.0f - (corner.z + float(k) * delta.z) / dist.x);
        r.y = delta.y * (1.0f - (corner.z + float(k) * delta.z) / dist.y);
        return r;
    }
    RTD_HD float stepVol(unsigned int k) const { return volConst + float(k) * volLin + float(k * k) * volSq; }  // :72
};
inline FillGeom makeFillGeom(float rRlScaleFact, const FromFan& t) {
    FillGeom p;
    p.rRlScale = rRlScaleFact;
    p.dist = t.dist; p.corner = t.fitf.offset; p.delta = t.fitf.delta;
    float a = fabsf(p.delta.x * p.delta.y * p.delta.z);
    p.volConst = a * (1.0f - p.corner.z / p.dist.x - p.corner.z / p.dist.y
                      + (p.corner.z * p.corner.z + p.delta.z * p.delta.z / 12.0f) / (p.dist.x * p.dist.y));
    p.volLin = a * p.delta.z * (-1.0f / p.dist.x - 1.0f / p.dist.y + 2.0f * p.corner.z / (p.dist.x * p.dist.y));
    p.volSq = a * p.delta.z * p.delta.z / (p.dist.x * p.dist.y);
    p.stepLength = fabsf(p.delta.z);   // initStepAndAirDiv :39
    return p;
}
RTD_HD Vec2 sigmaSqAirCoefs(float r0, int nozzle) {  // :74-83
    Vec2 r;
    if (nozzle) { r.x = 0.00270f / (r0 - 4.50f); r.y = -4.39f / (r0 - 3.86f); }
    else { r.x = 0.0f; r.y = 0.0f; }
    return r;
}

// ---- TransferParamStructDiv3 (transfer_param_struct_div3.cu:9-34) ----
struct TransferParams {
    Vec3 globalOffset, coefOffset, coefIdxI, coefIdxJ, inc, start;
    Vec2 normDist;

    RTD_HD void init(int i, int j) { start = float(i) * coefIdxI + float(j) * coefIdxJ + coefOffset; }
    RTD_HD Vec3 getFanIdx(int k) const {
        Vec3 r = start + float(k) * inc;
        r.x *= 1 + r.z / (normDist.x - r.z);
        r.y *= 1 + r.z / (normDist.y - r.z);
        return r + globalOffset;
    }
};
inline TransferParams makeTransferParams(const ToFan& t) {
    TransferParams p;
    Mat3 tT = transpose(t.iitg.m);
    Vec3 delta = t.ftfi.delta;
    p.coefIdxI = tT.r0 * delta;
    p.coefIdxJ = tT.r1 * delta;
    p.coefOffset = t.iitg.v * delta;
    p.globalOffset = t.ftfi.offset;
    p.inc = tT.r2 * delta;
    p.start = v3(0.0f, 0.0f, 0.0f);
    p.normDist.x = delta.z * t.dist.x;
    p.normDist.y = delta.z * t.dist.y;
    return p;
}

// ---- host search / interpolation helpers (vector_find.h, vector_interpolate.h), usable on device ----
RTD_HD int findFirstLargerOrdered(const float* list, int n, float value) {  // vector_find.h:60-82
    int upper = n - 1, lower = 0;
    if (list[n - 1] <= value) return upper;
    else if (list[0] > value) return 0;
    while (upper - lower > 1) {
        int pivot = (upper + lower) / 2;
        if (list[pivot] <= value) lower = pivot; else upper = pivot;
    }
    return lower + 1;
}
RTD_HD int findLastSmallerOrEqOrdered(const float* list, int n, float value) {  // vector_find.h:92-114
    int upper = n - 1, lower = 0;
    if (list[n - 1] <= value) return upper;
    else if (list[0] > value) return -1;
    while (upper - lower > 1) {
        int pivot = (upper + lower) / 2;
        if (list[pivot] <= value) lower = pivot; else upper = pivot;
    }
    return lower;
}
inline float findDecimalOrdered(const float* list, int n, float value) {  // vector_find.h:128-144
    if (value >= list[n - 1]) return float(n - 1);
    else if (value < list[0]) return 0.0f;
    unsigned int fl = (unsigned int)findLastSmallerOrEqOrdered(list, n, value);
    float corr = (value - list[fl]) / (list[fl + 1] - list[fl]);
    return float(fl) + corr;
}
inline float vectorInterpolate(const float* list, int n, float idx) {  // vector_interpolate.h:17-30
    if (idx <= 0.0f) return list[0];
    else if (idx >= float(n - 1)) return list[n - 1];
    float intPart;
    float decimals = std::modf(idx, &intPart);
    unsigned int fl = (unsigned int)intPart;
    float corr = (list[fl + 1] - list[fl]) * decimals;
    return list[fl] + corr;
}

constexpr int kSuperpTileX = 32;     // kernel_wrapper.cuh:27  (classification tile, NOT the wave width here)
constexpr int kSuperpTileY = 8;      // kernel_wrapper.cuh:28
constexpr int kMaxSuperpR = 32;      // kernel_wrapper.cuh:26
constexpr int kMinTilesInBatch = 16; // kernel_wrapper.cuh:29

inline int roundTo(int val, int multiple) { return ((val + multiple - 1) / multiple) * multiple; }  // kernel_wrapper.cu:45-48

}  // namespace rtd
