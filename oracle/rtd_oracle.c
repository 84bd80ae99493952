/*
 * rtd_oracle.c — CPU ORACLE of the pencil-beam proton dose hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference algorithm (ferdymercury/RayTraceDicom,
 * src/kernel_wrapper.cu:381-1369 and the files it calls). It exists to check the HIP engine and to
 * serve as the timed CPU baseline in bench.py. Nothing in the product path (raytracedicom_amd/,
 * include/) may call, link or import it; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * PARITY STATUS: the reference has no tests, no golden vectors and its device kernels cannot be
 * compiled in this environment (CUDA textures; no nvcc). So for the kernels this oracle is
 * "parity unpinned" by the reference; what IS pinned against the real reference code compiled
 * here (oracle/_ref, see oracle/Makefile) is: the LUT text parser (energy_reader.cpp), the host
 * search/interpolation helpers (vector_find.h, vector_interpolate.h) and the erf-difference
 * convolution weights (cpu_convolution_1d.cpp). See tests/test_oracle_golden.py.
 *
 * Sampler semantics restated from the CUDA texture unit (kernel_wrapper.cu:418-537): linear
 * filtering, un-normalised coordinates, "+0.5" texel-centre convention, BORDER(=0) for the CT and the
 * BEV dose, CLAMP for the LUTs. The reference adds 0.5 to every coordinate and the texture unit
 * subtracts it again; the restatement samples at the index-space position directly and uses exact
 * float weights (the hardware quantises weights to 8 fractional bits — not reproduced).
 *
 * The one transcendental of the sigma recurrence, __powf (kernel_wrapper.cu:282), is restated as rtd_pow_det
 * (include/rtd_detmath.h): IEEE-only arithmetic shared with the engine so that everything computed from it (the
 * radius class of every tile, the batch radii) is compared BIT-EXACTLY; its accuracy against a double-precision
 * pow is pinned by tests/test_oracle_kat.py.
 *
 * Every function cites the reference lines it follows. Arithmetic is written in the same operand
 * order as the reference and compiled with -ffp-contract=off so that the HIP kernels (written with
 * the same order) can be compared tightly.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/rtd.h"
#include "../include/rtd_detmath.h"   /* rtd_pow_det: the bit-reproducible stand-in for CUDA's __powf, shared with the engine */
#include "rtd_oracle.h"

/* ------------------------------------------------------------------------------------------ */
/* float3 helpers with the operator semantics of CUDA-samples helper_math.h (component-wise).  */

typedef struct { float x, y, z; } f3;
typedef struct { float x, y; } f2;

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 div3(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline f3 muls(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 smul(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
static inline f3 divs(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* Matrix3x3, row-major (matrix_3x3.cu). */
typedef struct { f3 r0, r1, r2; } m33;

static m33 m33_from(const float m[9]) {
    m33 r = { mk3(m[0], m[1], m[2]), mk3(m[3], m[4], m[5]), mk3(m[6], m[7], m[8]) };
    return r;
}
static m33 m33_transpose(m33 a) { /* matrix_3x3.cu:55-57 */
    m33 r = { mk3(a.r0.x, a.r1.x, a.r2.x), mk3(a.r0.y, a.r1.y, a.r2.y), mk3(a.r0.z, a.r1.z, a.r2.z) };
    return r;
}
static f3 m33_mulv(m33 m, f3 a) { return mk3(dot3(m.r0, a), dot3(m.r1, a), dot3(m.r2, a)); } /* :23-26 */
static m33 m33_mulm(m33 m1, m33 m) { /* matrix_3x3.cu:28-34 */
    m33 m2 = m33_transpose(m);
    m33 r = { mk3(dot3(m1.r0, m2.r0), dot3(m1.r0, m2.r1), dot3(m1.r0, m2.r2)),
              mk3(dot3(m1.r1, m2.r0), dot3(m1.r1, m2.r1), dot3(m1.r1, m2.r2)),
              mk3(dot3(m1.r2, m2.r0), dot3(m1.r2, m2.r1), dot3(m1.r2, m2.r2)) };
    return r;
}
static float m33_det(m33 a) { /* matrix_3x3.cu:42-44 */
    return (a.r0.x * (a.r1.y * a.r2.z - a.r1.z * a.r2.y) - a.r0.y * (a.r1.x * a.r2.z - a.r1.z * a.r2.x)
            + a.r0.z * (a.r1.x * a.r2.y - a.r1.y * a.r2.x));
}
static m33 m33_inverse(m33 a) { /* matrix_3x3.cu:46-53: adjugate * diag(1/det); 1./det is a double division */
    float oneOverDet = (float)(1.0 / (double)m33_det(a));
    m33 adj = { mk3(a.r1.y * a.r2.z - a.r1.z * a.r2.y, a.r0.z * a.r2.y - a.r0.y * a.r2.z, a.r0.y * a.r1.z - a.r0.z * a.r1.y),
                mk3(a.r1.z * a.r2.x - a.r1.x * a.r2.z, a.r0.x * a.r2.z - a.r0.z * a.r2.x, a.r0.z * a.r1.x - a.r0.x * a.r1.z),
                mk3(a.r1.x * a.r2.y - a.r1.y * a.r2.x, a.r0.y * a.r2.x - a.r0.x * a.r2.y, a.r0.x * a.r1.y - a.r0.y * a.r1.x) };
    /* "* oneOverDet" converts the scalar to a diagonal Matrix3x3 (matrix_3x3.cu:17-19) and multiplies. */
    m33 d = { mk3(oneOverDet, 0.0f, 0.0f), mk3(0.0f, oneOverDet, 0.0f), mk3(0.0f, 0.0f, oneOverDet) };
    return m33_mulm(adj, d);
}

typedef struct { m33 m; f3 v; } affine;
typedef struct { f3 delta, offset; } idxtr;

static affine affine_from(const rtd_affine* a) {
    affine r; r.m = m33_from(a->m); r.v = mk3(a->v[0], a->v[1], a->v[2]); return r;
}
static void affine_to(affine a, rtd_affine* o) {
    o->m[0] = a.m.r0.x; o->m[1] = a.m.r0.y; o->m[2] = a.m.r0.z;
    o->m[3] = a.m.r1.x; o->m[4] = a.m.r1.y; o->m[5] = a.m.r1.z;
    o->m[6] = a.m.r2.x; o->m[7] = a.m.r2.y; o->m[8] = a.m.r2.z;
    o->v[0] = a.v.x; o->v[1] = a.v.y; o->v[2] = a.v.z;
}
static idxtr idx_from(const rtd_idx_transform* t) {
    idxtr r; r.delta = mk3(t->delta[0], t->delta[1], t->delta[2]);
    r.offset = mk3(t->offset[0], t->offset[1], t->offset[2]); return r;
}
static f3 affine_point(affine a, f3 p) { return add3(m33_mulv(a.m, p), a.v); } /* float3_affine_transform.cu:14 */
static affine affine_inverse(affine a) { /* float3_affine_transform.cu:20-23 */
    affine r; r.m = m33_inverse(a.m); r.v = m33_mulv(m33_inverse(a.m), muls(a.v, -1.0f)); return r;
}
static affine affine_concat(affine t1, affine t2) { /* float3_affine_transform.cu:42-45: t1 then t2 */
    affine r; r.m = m33_mulm(t2.m, t1.m); r.v = add3(m33_mulv(t2.m, t1.v), t2.v); return r;
}
static f3 idx_point(idxtr t, f3 p) { return add3(mul3(p, t.delta), t.offset); } /* float3_idx_transform.cu:17 */
static idxtr idx_inverse(idxtr t) { /* float3_idx_transform.cu:19 */
    idxtr r; r.delta = div3(mk3(1.0f, 1.0f, 1.0f), t.delta);
    r.offset = div3(smul(-1.0f, t.offset), t.delta); return r;
}
static idxtr idx_shift(idxtr t, f3 s) { idxtr r = t; r.offset = add3(t.offset, s); return r; } /* :21 */

/* Float3FromFanTransform / Float3ToFanTransform (float3_from_fan_transform.cu:37-42, float3_to_fan_transform.cu:26-31). */
typedef struct { idxtr fitf; affine gtii; f2 dist; } fromfan;
typedef struct { affine iitg; f2 dist; idxtr ftfi; } tofan;

static f3 fromfan_point(const fromfan* t, f3 fanIdx) {
    f3 interm = idx_point(t->fitf, fanIdx);
    interm.x *= 1.0f - interm.z / t->dist.x;
    interm.y *= 1.0f - interm.z / t->dist.y;
    return affine_point(t->gtii, interm);
}
static f3 tofan_point(const tofan* t, f3 imIdx) {
    f3 interm = affine_point(t->iitg, imIdx);
    interm.x /= 1.0f - interm.z / t->dist.x;
    interm.y /= 1.0f - interm.z / t->dist.y;
    return idx_point(t->ftfi, interm);
}
static tofan fromfan_invert_and_shift(const fromfan* t, f3 shift) { /* float3_from_fan_transform.cu:27-30 */
    tofan r; r.iitg = affine_inverse(t->gtii); r.dist = t->dist;
    r.ftfi = idx_shift(idx_inverse(t->fitf), shift); return r;
}

/* ------------------------------------------------------------------------------------------ */
/* Host search / interpolation helpers (vector_find.h, vector_interpolate.h).                  */

float orc_find_max(const float* list, int n) { /* vector_find.h:19-30; caller guarantees n>0 */
    float m = list[0];
    for (int i = 1; i < n; ++i) if (list[i] > m) m = list[i];
    return m;
}
int orc_find_first_larger_ordered(const float* list, int n, float value) { /* vector_find.h:60-82 */
    int upper = n - 1, lower = 0;
    if (list[n - 1] <= value) return upper;
    else if (list[0] > value) return 0;
    while (upper - lower > 1) {
        int pivot = (upper + lower) / 2;
        if (list[pivot] <= value) lower = pivot; else upper = pivot;
    }
    return lower + 1;
}
int orc_find_last_smaller_or_eq_ordered(const float* list, int n, float value) { /* vector_find.h:92-114 */
    int upper = n - 1, lower = 0;
    if (list[n - 1] <= value) return upper;
    else if (list[0] > value) return -1;
    while (upper - lower > 1) {
        int pivot = (upper + lower) / 2;
        if (list[pivot] <= value) lower = pivot; else upper = pivot;
    }
    return lower;
}
float orc_find_decimal_ordered(const float* list, int n, float value) { /* vector_find.h:128-144 */
    if (value >= list[n - 1]) return (float)(n - 1);
    else if (value < list[0]) return 0.0f;
    unsigned int fl = (unsigned int)orc_find_last_smaller_or_eq_ordered(list, n, value);
    float corr = (value - list[fl]) / (list[fl + 1] - list[fl]);
    return (float)fl + corr;
}
float orc_vector_interpolate(const float* list, int n, float idx) { /* vector_interpolate.h:17-30 */
    if (idx <= 0.0f) return list[0];
    else if (idx >= (float)(n - 1)) return list[n - 1];
    float intPart;
    float decimals = modff(idx, &intPart);
    unsigned int fl = (unsigned int)intPart;
    float corr = (list[fl + 1] - list[fl]) * decimals;
    return list[fl] + corr;
}

/* ------------------------------------------------------------------------------------------ */
/* Samplers: software image of the texture fetches.                                           */

/* Optional emulation of the texture unit's fixed-point interpolation weights: CUDA stores the fractional position in
 * 9-bit fixed point with 8 fractional bits (CUDA C Programming Guide, "Linear Filtering"). 0 = exact float weights
 * (default, what the HIP engine implements); 8 = weights rounded to 1/256. Used only to QUANTIFY how far a real run
 * of the reference on NVIDIA hardware is expected to sit from the float-exact restatement. */
static int g_weight_bits = 0;
void orc_set_weight_bits(int bits) { g_weight_bits = bits; }
static inline float quant_w(float a) {
    if (g_weight_bits <= 0) return a;
    const float q = (float)(1 << g_weight_bits);
    return floorf(a * q + 0.5f) / q;
}
static inline float lerp_w(float a, float v0, float v1) { a = quant_w(a); return (1.0f - a) * v0 + a * v1; }

/* tex1D, linear, CLAMP (kernel_wrapper.cu:476-537). p = coordinate without the +0.5. */
static inline float sample1d_clamp(const float* t, int n, float p) {
    float fl = floorf(p);
    float a = p - fl;
    int i0 = (int)fl, i1 = i0 + 1;
    if (!(p >= 0.0f)) { i0 = 0; i1 = 0; a = 0.0f; }          /* also catches NaN like a clamped fetch */
    if (i0 > n - 1) i0 = n - 1;
    if (i1 > n - 1) i1 = n - 1;
    return lerp_w(a, t[i0], t[i1]);
}
/* tex2D, linear, CLAMP both axes (kernel_wrapper.cu:453-474): t[row][col], col = px, row = py. */
static inline float sample2d_clamp(const float* t, int ncol, int nrow, float px, float py) {
    float fx = floorf(px), fy = floorf(py);
    float ax = px - fx, ay = py - fy;
    int x0 = (int)fx, x1 = x0 + 1, y0 = (int)fy, y1 = y0 + 1;
    if (!(px >= 0.0f)) { x0 = 0; x1 = 0; ax = 0.0f; }
    if (!(py >= 0.0f)) { y0 = 0; y1 = 0; ay = 0.0f; }
    if (x0 > ncol - 1) x0 = ncol - 1;
    if (x1 > ncol - 1) x1 = ncol - 1;
    if (y0 > nrow - 1) y0 = nrow - 1;
    if (y1 > nrow - 1) y1 = nrow - 1;
    float r0 = lerp_w(ax, t[(size_t)y0 * ncol + x0], t[(size_t)y0 * ncol + x1]);
    float r1 = lerp_w(ax, t[(size_t)y1 * ncol + x0], t[(size_t)y1 * ncol + x1]);
    return lerp_w(ay, r0, r1);
}
/* tex3D, linear, BORDER (=0 outside) (kernel_wrapper.cu:420-451, 1107-1141). vol[z][y][x]. */
static inline float fetch3d_border(const float* vol, int nx, int ny, int nz, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return 0.0f;
    return vol[((size_t)z * ny + y) * nx + x];
}
static inline float sample3d_border(const float* vol, int nx, int ny, int nz, float px, float py, float pz) {
    /* positions far outside the volume (or NaN) are all-border */
    if (!(px > -1.0f && py > -1.0f && pz > -1.0f && px < (float)nx && py < (float)ny && pz < (float)nz)) return 0.0f;
    float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    float ax = px - fx, ay = py - fy, az = pz - fz;
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    float c00 = lerp_w(ax, fetch3d_border(vol, nx, ny, nz, x0, y0, z0), fetch3d_border(vol, nx, ny, nz, x0 + 1, y0, z0));
    float c10 = lerp_w(ax, fetch3d_border(vol, nx, ny, nz, x0, y0 + 1, z0), fetch3d_border(vol, nx, ny, nz, x0 + 1, y0 + 1, z0));
    float c01 = lerp_w(ax, fetch3d_border(vol, nx, ny, nz, x0, y0, z0 + 1), fetch3d_border(vol, nx, ny, nz, x0 + 1, y0, z0 + 1));
    float c11 = lerp_w(ax, fetch3d_border(vol, nx, ny, nz, x0, y0 + 1, z0 + 1), fetch3d_border(vol, nx, ny, nz, x0 + 1, y0 + 1, z0 + 1));
    float c0 = lerp_w(ay, c00, c10);
    float c1 = lerp_w(ay, c01, c11);
    return lerp_w(az, c0, c1);
}

/* float -> int conversion with the GPU's semantics (NaN -> 0, saturating), used for the tile radius. */
static inline int f2i_sat(float v) {
    if (v != v) return 0;
    if (v >= 2147483520.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

/* ------------------------------------------------------------------------------------------ */
/* Kernel parameter PODs.                                                                     */

/* DensityAndSpTracerParams (density_and_sp_tracer_params.cu:15-36). */
typedef struct {
    float densityScale, spScale; unsigned int steps;
    f3 coefOffset, coefIdxI, coefIdxJ, transl, corner, delta; f2 dist;
} tracer_params;

static tracer_params tracer_params_make(float densityScaleFact, float spScaleFact, unsigned int steps, const fromfan* t) {
    tracer_params p;
    p.densityScale = densityScaleFact; p.spScale = spScaleFact; p.steps = steps;
    p.dist = t->dist; p.corner = t->fitf.offset; p.delta = t->fitf.delta;
    m33 tT = m33_transpose(t->gtii.m);
    p.coefOffset = sub3(sub3(tT.r2, divs(muls(tT.r0, p.corner.x), p.dist.x)), divs(muls(tT.r1, p.corner.y), p.dist.y));
    p.coefIdxI = muls(tT.r0, p.delta.x);
    p.coefIdxJ = muls(tT.r1, p.delta.y);
    p.transl = add3(add3(add3(t->gtii.v, muls(tT.r2, p.corner.z)),
                         muls(muls(tT.r0, p.corner.x), (1.0f - p.corner.z / p.dist.x))),
                    muls(muls(tT.r1, p.corner.y), (1.0f - p.corner.z / p.dist.y)));
    return p;
}
static f3 tracer_get_start(const tracer_params* p, int i, int j) { /* :15 */
    return add3(add3(muls(smul((float)i, p->coefIdxI), (1.0f - p->corner.z / p->dist.x)),
                     muls(smul((float)j, p->coefIdxJ), (1.0f - p->corner.z / p->dist.y))), p->transl);
}
static f3 tracer_get_inc(const tracer_params* p, int i, int j) { /* :17 */
    return muls(sub3(sub3(p->coefOffset, divs(smul((float)i, p->coefIdxI), p->dist.x)),
                     divs(smul((float)j, p->coefIdxJ), p->dist.y)), p->delta.z);
}
static float tracer_step_len(const tracer_params* p, int i, int j) { /* :32-36 */
    float deltaX = (p->corner.x + (float)i * p->delta.x) / p->dist.x;
    float deltaY = (p->corner.y + (float)j * p->delta.y) / p->dist.y;
    return fabsf(p->delta.z) * sqrtf(1.0f + deltaX * deltaX + deltaY * deltaY);
}

/* FillIddAndSigmaParams (fill_idd_and_sigma_params.cu:12-83). */
typedef struct {
    float energyIdx, energyScaleFact, peakDepth, stepLength, sigmaSqAirLin, sigmaSqAirQuad, rRlScale;
    unsigned int first, afterLast;
    f3 corner, delta; f2 dist;
    float volConst, volLin, volSq;
} fill_params;

static f2 sigma_sq_air_coefs(float r0, int nozzle) { /* :74-83 */
    f2 r;
    if (nozzle) { r.x = 0.00270f / (r0 - 4.50f); r.y = -4.39f / (r0 - 3.86f); }
    else { r.x = 0.0f; r.y = 0.0f; }
    return r;
}
static fill_params fill_params_make(float energyIdx, float energyScaleFact, float peakDepth, float rRlScaleFact,
                                    unsigned int first, unsigned int afterLast, const fromfan* t, int nozzle) {
    fill_params p;
    p.energyIdx = energyIdx; p.energyScaleFact = energyScaleFact; p.peakDepth = peakDepth; p.rRlScale = rRlScaleFact;
    p.first = first; p.afterLast = afterLast;
    p.dist = t->dist; p.corner = t->fitf.offset; p.delta = t->fitf.delta;
    float a = fabsf(p.delta.x * p.delta.y * p.delta.z);
    p.volConst = a * (1.0f - p.corner.z / p.dist.x - p.corner.z / p.dist.y
                      + (p.corner.z * p.corner.z + p.delta.z * p.delta.z / 12.0f) / (p.dist.x * p.dist.y));
    p.volLin = a * p.delta.z * (-1.0f / p.dist.x - 1.0f / p.dist.y + 2.0f * p.corner.z / (p.dist.x * p.dist.y));
    p.volSq = a * p.delta.z * p.delta.z / (p.dist.x * p.dist.y);
    /* initStepAndAirDiv (:28-40) */
    float relStepLenSq = 1.0f;
    f2 c = sigma_sq_air_coefs(peakDepth, nozzle);
    p.sigmaSqAirQuad = c.x * relStepLenSq * p.delta.z * p.delta.z;
    float zDist = p.corner.z;
    p.sigmaSqAirLin = 2.0f * c.x * relStepLenSq * p.delta.z * zDist + c.y * p.delta.z;
    p.stepLength = fabsf(p.delta.z);
    return p;
}
static inline f2 fill_voxel_width(const fill_params* p, unsigned int k) { /* :42-46 */
    f2 r;
    r.x = p->delta.x * (1.0f - (p->corner.z + (float)k * p->delta.z) / p->dist.x);
    r.y = p->delta.y * (1.0f - (p->corner.z + (float)k * p->delta.z) / p->dist.y);
    return r;
}
static inline float fill_step_vol(const fill_params* p, unsigned int k) { /* :72 */
    return p->volConst + (float)k * p->volLin + (float)(k * k) * p->volSq;
}

/* TransferParamStructDiv3 (transfer_param_struct_div3.cu:9-34). */
typedef struct { f3 globalOffset, coefOffset, coefIdxI, coefIdxJ, inc, start; f2 normDist; } transfer_params;

static transfer_params transfer_params_make(const tofan* t) {
    transfer_params p;
    m33 tT = m33_transpose(t->iitg.m);
    f3 delta = t->ftfi.delta;
    p.coefIdxI = mul3(tT.r0, delta);
    p.coefIdxJ = mul3(tT.r1, delta);
    p.coefOffset = mul3(t->iitg.v, delta);
    p.globalOffset = t->ftfi.offset;
    p.inc = mul3(tT.r2, delta);
    p.start = mk3(0.0f, 0.0f, 0.0f);
    p.normDist.x = delta.z * t->dist.x;
    p.normDist.y = delta.z * t->dist.y;
    return p;
}
static inline void transfer_init(transfer_params* p, int i, int j) { /* :22-25 */
    p->start = add3(add3(smul((float)i, p->coefIdxI), smul((float)j, p->coefIdxJ)), p->coefOffset);
}
static inline f3 transfer_get_fan_idx(const transfer_params* p, int k) { /* :27-34 */
    f3 r = add3(p->start, smul((float)k, p->inc));
    r.x *= 1 + r.z / (p->normDist.x - r.z);
    r.y *= 1 + r.z / (p->normDist.y - r.z);
    return add3(r, p->globalOffset);
}

/* ------------------------------------------------------------------------------------------ */
/* Exposed param probes (used by tests to check invariants such as the main.cu:220-238 probe). */

static fromfan make_fromfan(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx) {
    fromfan t; t.fitf = idx_from(fanIdxToFan); t.gtii = affine_from(gantryToIdx);
    t.dist.x = sourceDist[0]; t.dist.y = sourceDist[1]; return t;
}

void orc_affine_inverse(const rtd_affine* in, rtd_affine* out) { affine_to(affine_inverse(affine_from(in)), out); }
void orc_affine_concat(const rtd_affine* t1, const rtd_affine* t2, rtd_affine* out) {
    affine_to(affine_concat(affine_from(t1), affine_from(t2)), out);
}
void orc_from_fan_point(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                        const float in[3], float out[3]) {
    fromfan t = make_fromfan(fanIdxToFan, sourceDist, gantryToIdx);
    f3 r = fromfan_point(&t, mk3(in[0], in[1], in[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_to_fan_point(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                      const float shift[3], const float in[3], float out[3]) {
    fromfan t = make_fromfan(fanIdxToFan, sourceDist, gantryToIdx);
    tofan inv = fromfan_invert_and_shift(&t, mk3(shift[0], shift[1], shift[2]));
    f3 r = tofan_point(&inv, mk3(in[0], in[1], in[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_transfer_fan_idx(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                          const float shift[3], int i, int j, int k, float out[3]) {
    fromfan t = make_fromfan(fanIdxToFan, sourceDist, gantryToIdx);
    tofan inv = fromfan_invert_and_shift(&t, mk3(shift[0], shift[1], shift[2]));
    transfer_params p = transfer_params_make(&inv);
    transfer_init(&p, i, j);
    f3 r = transfer_get_fan_idx(&p, k);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_tracer_probe(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                      int i, int j, float start[3], float inc[3], float* stepLen) {
    fromfan t = make_fromfan(fanIdxToFan, sourceDist, gantryToIdx);
    tracer_params p = tracer_params_make(1.0f, 1.0f, 1, &t);
    f3 s = tracer_get_start(&p, i, j), d = tracer_get_inc(&p, i, j);
    start[0] = s.x; start[1] = s.y; start[2] = s.z; inc[0] = d.x; inc[1] = d.y; inc[2] = d.z;
    *stepLen = tracer_step_len(&p, i, j);
}
void orc_fill_probe(const rtd_idx_transform* fanIdxToFan, const float sourceDist[2], const rtd_affine* gantryToIdx,
                    float peakDepth, int nozzle, unsigned int k, float* stepVol, float voxelWidth[2],
                    float* sigmaSqAirLin, float* sigmaSqAirQuad) {
    fromfan t = make_fromfan(fanIdxToFan, sourceDist, gantryToIdx);
    fill_params p = fill_params_make(0.0f, 1.0f, peakDepth, 1.0f, 0, 0, &t, nozzle);
    *stepVol = fill_step_vol(&p, k);
    f2 w = fill_voxel_width(&p, k); voxelWidth[0] = w.x; voxelWidth[1] = w.y;
    *sigmaSqAirLin = p.sigmaSqAirLin; *sigmaSqAirQuad = p.sigmaSqAirQuad;
}
float orc_sample1d(const float* t, int n, float p) { return sample1d_clamp(t, n, p); }
float orc_sample2d(const float* t, int ncol, int nrow, float px, float py) { return sample2d_clamp(t, ncol, nrow, px, py); }
float orc_sample3d(const float* vol, int nx, int ny, int nz, float px, float py, float pz) {
    return sample3d_border(vol, nx, ny, nz, px, py, pz);
}

/* Erf-difference weights of the kernel superposition (kernel_wrapper.cuh:459-467); n = rad+1 entries. */
void orc_erf_diffs(float rSigmaEff, int rad, float* out) {
    float erfNew = erff(rSigmaEff * 0.5f);
    float erfOld = -erfNew;
    for (int i = 0; i <= rad; ++i) {
        out[i] = 0.5f * (erfNew - erfOld);
        erfOld = erfNew;
        erfNew = erff(rSigmaEff * ((float)i + 1.5f));
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Stage 1: fillBevDensityAndSp (kernel_wrapper.cu:130-187).                                   */

/* CT footprint of the tracer = number of DISTINCT CT voxels its trilinear samples read (the N_fp of the algorithmic-byte
 * model). While a map is armed (orc_footprint_start) every sample marks its in-range corner voxels. */
static unsigned char* g_touch = NULL;
static size_t g_touch_n = 0;
void orc_footprint_start(size_t nVoxels) { free(g_touch); g_touch = (unsigned char*)calloc(nVoxels ? nVoxels : 1, 1); g_touch_n = g_touch ? nVoxels : 0; }
long long orc_footprint_stop(void) {
    long long c = 0;
    for (size_t i = 0; i < g_touch_n; ++i) c += g_touch[i];
    free(g_touch); g_touch = NULL; g_touch_n = 0;
    return c;
}
static inline void touch_ct(int nx, int ny, int nz, float px, float py, float pz) {
    if (!(px > -1.0f && py > -1.0f && pz > -1.0f && px < (float)nx && py < (float)ny && pz < (float)nz)) return;
    const int x0 = (int)floorf(px), y0 = (int)floorf(py), z0 = (int)floorf(pz);
    for (int dz = 0; dz < 2; ++dz) for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {
        const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
        if (x >= 0 && y >= 0 && z >= 0 && x < nx && y < ny && z < nz) g_touch[((size_t)z * ny + y) * nx + x] = 1;   /* (benign race: all writers store 1) */
    }
}

static void stage_trace(const float* ct, const int ctd[3], const rtd_luts* l, const tracer_params* tp,
                        int W, int H, float* bevDensity, float* bevCumulSp, int* firstInside, int* firstOutside) {
    const size_t memStep = (size_t)W * H;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            size_t idx = (size_t)y * W + x;
            f3 pos = tracer_get_start(tp, x, y);   /* the +0.5 of :142 is the texel-centre convention */
            f3 step = tracer_get_inc(tp, x, y);
            float stepLen = tracer_step_len(tp, x, y);
            float cumulSp = 0.0f, cumulHuPlus1000 = 0.0f;
            int beforeFirstInside = -1, lastInside = -1;
            for (unsigned int i = 0; i < tp->steps; ++i) {
                float huPlus1000 = sample3d_border(ct, ctd[0], ctd[1], ctd[2], pos.x, pos.y, pos.z);
                if (g_touch && g_touch_n == (size_t)ctd[0] * ctd[1] * ctd[2]) touch_ct(ctd[0], ctd[1], ctd[2], pos.x, pos.y, pos.z);
                cumulHuPlus1000 += huPlus1000;
                bevDensity[idx] = sample1d_clamp(l->density_vector, l->n_density_samples, huPlus1000 * tp->densityScale);
                cumulSp += stepLen * sample1d_clamp(l->sp_vector, l->n_sp_samples, huPlus1000 * tp->spScale);
                if (cumulHuPlus1000 < 150.0f) beforeFirstInside = (int)i;
                if (huPlus1000 > 150.0f) lastInside = (int)i;
                bevCumulSp[idx] = cumulSp;
                idx += memStep;
                pos = add3(pos, step);
            }
            firstInside[(size_t)y * W + x] = beforeFirstInside + 1;
            firstOutside[(size_t)y * W + x] = lastInside + 1;
        }
    }
}

/* Stage 2: gpuConvolution2D (gpu_convolution_2d.cu:16-71). rsqrtf(2.0f) is written 1/sqrtf(2); the device erff is
 * rtd_erf_det (include/rtd_detmath.h), the same bits as the engine under test evaluates. */
static void stage_conv2d(const float* in, float* interm, float* out, const f2* sigmas, const unsigned int inDims[3],
                         const unsigned int outDims[3], f3 spotDelta, f3 spotOffset, f3 rayDelta, f3 rayOffset,
                         f2 pxSpMult, float convCut) {
    const float inOutDeltaX = spotDelta.x / rayDelta.x, inOutDeltaY = spotDelta.y / rayDelta.y;
    const float inOutOffsetX = (spotOffset.x - rayOffset.x) / rayDelta.x, inOutOffsetY = (spotOffset.y - rayOffset.y) / rayDelta.y;
    const float pixelSpX = rayDelta.x * pxSpMult.x, pixelSpY = rayDelta.y * pxSpMult.y;
    const int inW = (int)inDims[0], inH = (int)inDims[1], L = (int)inDims[2], outW = (int)outDims[0], outH = (int)outDims[1];
    for (int z = 0; z < L; ++z) {
        /* xConvGathResampGpu (:16-35) */
        for (int idxY = 0; idxY < inH; ++idxY) for (int outIdxX = 0; outIdxX < outW; ++outIdxX) {
            float res = 0.0f;
            float sigmaEff = sigmas[z].x / pixelSpX;
            float rSigmaEff = (1.0f / sqrtf(2.0f)) / sigmaEff;
            int cur = f2i_sat(ceilf(((float)outIdxX - (convCut * sigmaEff + 0.5f) - inOutOffsetX) / inOutDeltaX));
            if (cur < 0) cur = 0;   /* spots left of the map contribute nothing; bounds the loop, same result */
            float dist = (float)cur * inOutDeltaX + inOutOffsetX - (float)outIdxX;
            while (dist < (convCut * sigmaEff + 0.5f) && cur < inW) {
                if (cur >= 0 && cur < inW)
                    res += 0.5f * (rtd_erf_det((dist + 0.5f) * rSigmaEff) - rtd_erf_det((dist - 0.5f) * rSigmaEff))
                           * in[(size_t)z * inW * inH + (size_t)idxY * inW + cur];
                ++cur;
                dist = (float)cur * inOutDeltaX + inOutOffsetX - (float)outIdxX;
            }
            interm[(size_t)z * outW * inH + (size_t)idxY * outW + outIdxX] = res;
        }
        /* yConvGathResampGpu (:37-59) */
        for (int outIdxY = 0; outIdxY < outH; ++outIdxY) for (int idxX = 0; idxX < outW; ++idxX) {
            float res = 0.0f;
            float sigmaEff = sigmas[z].y / pixelSpY;
            float rSigmaEff = (1.0f / sqrtf(2.0f)) / sigmaEff;
            int cur = f2i_sat(ceilf(((float)outIdxY - (convCut * sigmaEff + 0.5f) - inOutOffsetY) / inOutDeltaY));
            if (cur < 0) cur = 0;
            float dist = (float)cur * inOutDeltaY + inOutOffsetY - (float)outIdxY;
            while (dist < (convCut * sigmaEff + 0.5f) && cur < inH) {
                if (cur >= 0 && cur < inH)
                    res += 0.5f * (rtd_erf_det((dist + 0.5f) * rSigmaEff) - rtd_erf_det((dist - 0.5f) * rSigmaEff))
                           * interm[(size_t)z * outW * inH + (size_t)cur * outW + idxX];
                ++cur;
                dist = (float)cur * inOutDeltaY + inOutOffsetY - (float)outIdxY;
            }
            out[(size_t)z * outW * outH + (size_t)outIdxY * outW + idxX] = res;
        }
    }
}

/* Stage 3: fillIddAndSigma without NUCLEAR_CORR (kernel_wrapper.cu:190-379). */
/* The NUCLEAR_CORR arguments of fillIddAndSigma (kernel_wrapper.cu:190-198): arrays on the nuclear (spot-resolution) grid. */
typedef struct {
    float* bevNucIdd; float* bevNucRSigmaEff;      /* [S][nucH][nucW], but see nucMemStep */
    const float* nucRayWeights;                    /* this layer's padded spot weights [nucH][nucW] (extendAndPadd, :51-66) */
    const int* nucIdcs;                            /* [H][W] index of the ray's spot on the nuclear grid, -1 if none (:878-892) */
    float spotDist, entrySigmaSq;                  /* FillIddAndSigmaParams::getSpotDist / getEntrySigmaSq */
    unsigned int nucMemStep;                       /* the reference passes 0 (:925, 7th constructor argument) */
} nuc_fill;

static void stage_fill(const float* bevDensity, const float* bevCumulSp, float* bevIdd, float* bevRSigmaEff,
                       const float* rayWeights, const int* firstInside, const int* firstOutside, int* firstPassive,
                       const fill_params* pp, const rtd_luts* l, const rtd_options* opt, int W, int H, const nuc_fill* nuc) {
    const size_t memStep = (size_t)W * H;
    const fill_params params = *pp;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        size_t idx = (size_t)y * W + x;
        int beamLive = 1;
        const int firstIn = firstInside[idx];
        int fo = firstOutside[idx], al = (int)params.afterLast;
        unsigned int afterLast = (unsigned int)(fo < al ? fo : al);
        const float rayWeight = rayWeights[idx];
        if (rayWeight < opt->ray_weight_cutoff || afterLast < params.first) { beamLive = 0; afterLast = 0; }

        float res = 0.0f, rSigmaEff = 0.0f, cumulSp, cumulSpOld = 0.0f, cumulDose, cumulDoseOld = 0.0f;
        const float pInv = 0.5649718f, eCoef = 8.639415f, sqrt2 = 1.41421356f;
        /* E_s^2 and the empirical widening per NUCLEAR_CORR variant (:228-245; "CORRECT ALL THESE" in the reference) */
        float eRefSq = 198.81f, sigmaDelta = 0.21f;
        if (opt->nuclear_corr == RTD_NUC_SOUKUP) { eRefSq = 190.44f; sigmaDelta = 0.0f; }
        else if (opt->nuclear_corr == RTD_NUC_FLUKA) { eRefSq = 216.09f; sigmaDelta = 0.08f; }
        else if (opt->nuclear_corr == RTD_NUC_GAUSS_FIT) { eRefSq = 169.00f; sigmaDelta = 0.06f; }
        float incScat = 0.0f, incincScat = 0.0f;
        float incDiv = params.sigmaSqAirLin + (2.0f * (float)params.first - 1.0f) * params.sigmaSqAirQuad;
        float sigmaSq = -incDiv;
        /* :253-263 */
        float nucRes = 0.0f, nucRSigmaEff = 0.0f, nucRayWeight = 0.0f;
        long long nucIdx = -1;
        if (nuc) {
            nucIdx = nuc->nucIdcs[idx];
            if (nucIdx >= 0) nucRayWeight = nuc->nucRayWeights[nucIdx];
            nucIdx += (long long)params.first * nuc->nucMemStep;
        }

        idx += (size_t)params.first * memStep;
        for (unsigned int stepNo = params.first; stepNo < params.afterLast; ++stepNo) {
            if (beamLive) {
                cumulSp = bevCumulSp[idx];
                cumulDose = sample2d_clamp(l->cidd_matrix, l->n_energy_samples, l->n_energies,
                                           cumulSp * params.energyScaleFact, params.energyIdx);
                float density = bevDensity[idx];
                if (cumulSp < params.peakDepth) {
                    /* __powf (kernel_wrapper.cu:282) is a hardware approximation no other machine reproduces; the
                     * restatement uses rtd_pow_det (include/rtd_detmath.h, <= 2.7e-7 relative, i.e. tighter than the
                     * intrinsic): same bits here and on the GPU, so the radius classes below can be compared exactly */
                    float resE = eCoef * rtd_pow_det(params.peakDepth - 0.5f * (cumulSp + cumulSpOld), pInv);
                    float betaP = resE + 938.3f - 938.3f * 938.3f / (resE + 938.3f);
                    float rRl = density * sample1d_clamp(l->rrl_vector, l->n_rrl_samples, density * params.rRlScale);
                    float thetaSq = eRefSq / (betaP * betaP) * params.stepLength * rRl;
                    sigmaSq += incScat + incDiv;
                    incincScat += 2.0f * thetaSq * params.stepLength * params.stepLength;
                    incScat += incincScat;
                    incDiv += 2.0f * params.sigmaSqAirQuad;
                } else {
                    if (opt->nuclear_corr != RTD_NUC_GAUSS_FIT)                       /* :300-302 */
                        sigmaSq -= 1.5f * (incScat + incDiv) * density;
                }
                f2 vw = fill_voxel_width(&params, stepNo);
                rSigmaEff = 0.5f * (vw.x + vw.y) / (sqrt2 * (sqrtf(sigmaSq) + sigmaDelta));
                if (cumulSp > params.peakDepth * opt->bp_depth_cutoff || stepNo == afterLast) {
                    beamLive = 0; afterLast = stepNo;
                }
                float mass = opt->dose_to_water ? (cumulSp - cumulSpOld) * fill_step_vol(&params, stepNo)
                                                : density * fill_step_vol(&params, stepNo);
                if (!nuc) {
                    if (mass > 1e-2f) res = rayWeight * (cumulDose - cumulDoseOld) / mass;
                } else {                                                              /* :320-341 */
                    const float px = 0.5f * (cumulSp + cumulSpOld) * params.energyScaleFact;
                    if (mass > 1e-2f) {
                        float nucWeight = sample2d_clamp(l->nuc_weight_matrix, l->n_energy_samples, l->n_energies, px, params.energyIdx);
                        res = (1.0f - nucWeight) * rayWeight * (cumulDose - cumulDoseOld) / mass;
                        nucRes = nucWeight * nucRayWeight * (cumulDose - cumulDoseOld) / (mass * nuc->spotDist * nuc->spotDist);
                    }
                    if (nucIdx >= 0) {
                        float nucSqSigma = sample2d_clamp(l->nuc_sq_sigma_matrix, l->n_energy_samples, l->n_energies, px, params.energyIdx);
                        nucRSigmaEff = 0.5f * nuc->spotDist * (vw.x + vw.y) / (sqrt2 * sqrtf(sigmaSq + nucSqSigma + nuc->entrySigmaSq));
                    }
                }
                cumulSpOld = cumulSp;
                cumulDoseOld = cumulDose;
            }
            if (!beamLive || (int)stepNo < (firstIn - 1)) { res = 0.0f; rSigmaEff = INFINITY; nucRes = 0.0f; nucRSigmaEff = INFINITY; }
            bevIdd[idx] = res;
            bevRSigmaEff[idx] = rSigmaEff;
            if (nuc) {                                                                /* :367-374 */
                if (nucIdx >= 0) { nuc->bevNucIdd[nucIdx] = nucRes; nuc->bevNucRSigmaEff[nucIdx] = nucRSigmaEff; }
                nucIdx += nuc->nucMemStep;
            }
            idx += memStep;
        }
        firstPassive[(size_t)y * W + x] = (int)afterLast;
    }
}

#define SUPERP_TILE_X 32   /* kernel_wrapper.cuh:27 */
#define SUPERP_TILE_Y 8    /* kernel_wrapper.cuh:28 */
#define MAX_SUPERP_R 32    /* kernel_wrapper.cuh:26 */
#define MIN_TILES_IN_BATCH 16 /* kernel_wrapper.cuh:29 */

/* Stage 4: tileRadCalc (kernel_wrapper.cuh:256-313): per (step, tile) radius class. 0xFF = outside range. */
static void stage_tile_radius(const float* rs, int W, int H, int first, int layerFirstPassive, float ksCut,
                              uint8_t* tileRad /* [S][tilesY][tilesX] */, int S, int ctrs[MAX_SUPERP_R + 2]) {
    const int tx = W / SUPERP_TILE_X, ty = H / SUPERP_TILE_Y;
    memset(tileRad, 0xFF, (size_t)S * tx * ty);
    for (int i = 0; i < MAX_SUPERP_R + 2; ++i) ctrs[i] = 0;
    for (int k = first; k < layerFirstPassive; ++k) for (int by = 0; by < ty; ++by) for (int bx = 0; bx < tx; ++bx) {
        const float* base = rs + (size_t)k * W * H + (size_t)(by * SUPERP_TILE_Y) * W + bx * SUPERP_TILE_X;
        float minVal = base[0];
        for (int r = 0; r < SUPERP_TILE_Y; ++r) for (int c = 0; c < SUPERP_TILE_X; ++c) {
            float t = base[(size_t)r * W + c];
            if (t < minVal) minVal = t;
        }
        int rad = f2i_sat(ksCut / (sqrtf(2.0f) * minVal) + 0.5f);
        if (rad > MAX_SUPERP_R + 1) rad = MAX_SUPERP_R + 1;
        if (rad < 0) rad = 0; /* unreachable for finite inputs; keeps the table index valid */
        tileRad[((size_t)k * ty + by) * tx + bx] = (uint8_t)rad;
        ctrs[rad] += 1;
    }
}

float orc_pow_det(float x, float y) { return rtd_pow_det(x, y); }   /* for the known-answer test of the shared routine */
float orc_erf_det(float x) { return rtd_erf_det(x); }

/* Host batching of radii (kernel_wrapper.cu:966-976): effRad[rad] = template radius of the launch that
 * serves tiles of radius rad (kernel_wrapper.cuh:443-448). Returns layerMaxPrimSuperpR. */
int orc_batch_radii(const int ctrs[MAX_SUPERP_R + 2], int effRad[MAX_SUPERP_R + 2]) {
    int layerMax = 0;
    for (int i = 0; i < MAX_SUPERP_R + 2; ++i) if (ctrs[i] > 0) layerMax = i;
    for (int i = 0; i < MAX_SUPERP_R + 2; ++i) effRad[i] = i;
    if (layerMax > MAX_SUPERP_R) return layerMax; /* overflow: caller raises */
    int rec = layerMax;
    int batched[MAX_SUPERP_R + 1];
    for (int i = 0; i <= MAX_SUPERP_R; ++i) batched[i] = 0;
    batched[0] = ctrs[0];
    effRad[0] = 0;
    for (int rad = layerMax; rad > 0; --rad) {
        batched[rec] += ctrs[rad];
        effRad[rad] = rec;
        if (batched[rec] >= MIN_TILES_IN_BATCH) rec = rad - 1;
    }
    return layerMax;
}

/* Stage 5: kernelSuperposition<rad> (kernel_wrapper.cuh:432-489) for every classified tile of one layer.
 * bev: padded [Z][H+64][W+64]. The shared-memory tile and the accumulation order inside a tile follow
 * the kernel; tiles are flushed into bev in (k, tileY, tileX) order (the reference's atomicAdd order is
 * not defined). Parallel over k: slices of bev are independent. */
static void stage_superposition(const float* idd, const float* rs, float* bev, int W, int H, const uint8_t* tileRad,
                                const int effRad[MAX_SUPERP_R + 2], int first, int layerFirstPassive) {
    const int tx = W / SUPERP_TILE_X, ty = H / SUPERP_TILE_Y;
    const int outPitch = W + 2 * MAX_SUPERP_R;
    const size_t outSlice = (size_t)outPitch * (H + 2 * MAX_SUPERP_R);
#pragma omp parallel
    {
        float* tile = (float*)malloc(sizeof(float) * (SUPERP_TILE_X + 2 * MAX_SUPERP_R) * (SUPERP_TILE_Y + 2 * MAX_SUPERP_R));
#pragma omp for schedule(dynamic, 1)
        for (int k = first; k < layerFirstPassive; ++k) for (int by = 0; by < ty; ++by) for (int bx = 0; bx < tx; ++bx) {
            int ownRad = tileRad[((size_t)k * ty + by) * tx + bx];
            if (ownRad == 0xFF || ownRad > MAX_SUPERP_R) continue;
            const int rad = effRad[ownRad];
            const int tw = SUPERP_TILE_X + 2 * rad, th = SUPERP_TILE_Y + 2 * rad;
            for (int i = 0; i < tw * th; ++i) tile[i] = 0.0f;
            const size_t inBase = (size_t)k * W * H + (size_t)(by * SUPERP_TILE_Y) * W + bx * SUPERP_TILE_X;
            /* The kernel runs the 8 rows concurrently (one warp-row each) and synchronises per i, so the
             * adds into one tile element arrive ordered by i; we keep that order: loop i outermost. */
            /* __syncthreads_or(dose>0) (:456) is block-wide: the tile is processed iff ANY of its 256 voxels has dose>0. */
            int tileActive = 0;
            float dose[SUPERP_TILE_Y][SUPERP_TILE_X];
            float ed[SUPERP_TILE_Y][SUPERP_TILE_X][MAX_SUPERP_R + 1];
            for (int row = 0; row < SUPERP_TILE_Y; ++row) for (int c = 0; c < SUPERP_TILE_X; ++c) {
                dose[row][c] = idd[inBase + (size_t)row * W + c];
                if (dose[row][c] > 0.0f) tileActive = 1;
            }
            if (tileActive) {
                for (int row = 0; row < SUPERP_TILE_Y; ++row) for (int c = 0; c < SUPERP_TILE_X; ++c)
                    orc_erf_diffs(rs[inBase + (size_t)row * W + c], rad, ed[row][c]);
                for (int i = 0; i < 2 * rad + 1; ++i) for (int row = 0; row < SUPERP_TILE_Y; ++row)
                    for (int j = 0; j < 2 * rad + 1; ++j) for (int c = 0; c < SUPERP_TILE_X; ++c)
                        tile[(row + i) * tw + c + j] += dose[row][c] * ed[row][c][abs(rad - i)] * ed[row][c][abs(rad - j)];
            }
            /* flush (:480-488) */
            const size_t outBase = (size_t)k * outSlice + (size_t)(by * SUPERP_TILE_Y) * outPitch + bx * SUPERP_TILE_X;
            for (int row = -rad + MAX_SUPERP_R; row < SUPERP_TILE_Y + rad + MAX_SUPERP_R; ++row)
                for (int col = -rad + MAX_SUPERP_R; col < SUPERP_TILE_X + rad + MAX_SUPERP_R; ++col)
                    bev[outBase + (size_t)row * outPitch + col] += tile[(row + rad - MAX_SUPERP_R) * tw + col + rad - MAX_SUPERP_R];
        }
        free(tile);
    }
}

/* Stage 6: primTransfDiv (kernel_wrapper.cu:69-97) over the bounding box. slab = bev + first*slice,
 * slab dims (W+64, H+64, calcPassive-first) with BORDER addressing (kernel_wrapper.cu:1107-1141). */
static void stage_transfer(float* dose, const unsigned int doseDims[3], const transfer_params* tp0, const int minIdx[3],
                           const int maxIdx[3], const float* slab, int sx, int sy, int sz) {
    /* The launch (kernel_wrapper.cu:1209-1210) rounds the grid up to whole 32 x 8 blocks in x AND y; the extra threads are masked
     * only by x < doseDims.x && y < doseDims.y (:80), so both axes run past maxIdx up to the block edge; z stops at maxIdx.z. */
    const int yEnd = minIdx[1] + ((maxIdx[1] - minIdx[1] + 1 + 7) / 8) * 8;
#pragma omp parallel for schedule(static)
    for (int y = minIdx[1]; y < yEnd; ++y) {
        if (y < 0 || y >= (int)doseDims[1]) continue;
        int xEnd = minIdx[0] + ((maxIdx[0] - minIdx[0] + 1 + 31) / 32) * 32;
        for (int x = minIdx[0]; x < xEnd; ++x) {
            if (x < 0 || x >= (int)doseDims[0]) continue;
            transfer_params p = *tp0;
            transfer_init(&p, x, y);
            float* res = dose + (size_t)minIdx[2] * doseDims[0] * doseDims[1] + (size_t)y * doseDims[0] + x;
            for (int z = minIdx[2]; z <= maxIdx[2]; ++z) {
                f3 pos = transfer_get_fan_idx(&p, z);
                float tmp = sample3d_border(slab, sx, sy, sz, pos.x, pos.y, pos.z);
                if (tmp > 0.0f) *res += tmp;
                res += (size_t)doseDims[0] * doseDims[1];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Orchestration of one field: the body of the beam loop (kernel_wrapper.cu:601-1312).         */

static int round_to(int val, int multiple) { return ((val + multiple - 1) / multiple) * multiple; } /* :45-48 */

struct orc_field_s {
    int W, H, L, S;
    float* density; float* wepl; int* firstInside; int* firstOutside; float* weplMin;
    float* rayWeights; float* idd; float* rsigma; int* firstPassive; uint8_t* tileRad; int* effRad;
    float* bev; float* layerPlan;
    int keepLayers;
    rtd_field_info info;
    char err[256];
};

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_field_free(orc_field f) {
    if (!f) return;
    free(f->density); free(f->wepl); free(f->firstInside); free(f->firstOutside); free(f->weplMin);
    free(f->rayWeights); free(f->idd); free(f->rsigma); free(f->firstPassive); free(f->tileRad); free(f->effRad);
    free(f->bev); free(f->layerPlan);
    free(f);
}

/* Geometry of the ray grid (kernel_wrapper.cu:612-663); shared with tests through orc_ray_grid. */
static int ray_grid(const rtd_beam* b, const rtd_options* opt, unsigned int rayDims[3], float rayRes[3], float rayOffset[3]) {
    if (b->n_layers == 0) return RTD_ERR_INVALID_ARG;
    float* xs = (float*)malloc(sizeof(float) * b->n_layers), *ys = (float*)malloc(sizeof(float) * b->n_layers);
    for (unsigned int i = 0; i < b->n_layers; ++i) { xs[i] = b->spot_sigmas[2 * i]; ys[i] = b->spot_sigmas[2 * i + 1]; }
    const float maxSx = orc_find_max(xs, (int)b->n_layers), maxSy = orc_find_max(ys, (int)b->n_layers);
    free(xs); free(ys);
    const idxtr sitg = idx_from(&b->spot_idx_to_gantry);
    const f3 res = mk3(b->ray_spacing[0], b->ray_spacing[1], sitg.delta.z);
    const float cc = opt->conv_sigma_cutoff;
    int lSteps = (int)ceilf((sitg.offset.x - (cc * maxSx + 0.5f * res.x)) / res.x);
    int bSteps = (int)ceilf((sitg.offset.y - (cc * maxSy + 0.5f * res.y)) / res.y);
    int rSteps = (int)floorf(((float)(b->spot_nx - 1) * sitg.delta.x + sitg.offset.x + (cc * maxSx + 0.5f * res.x)) / res.x);
    int tSteps = (int)floorf(((float)(b->spot_ny - 1) * sitg.delta.y + sitg.offset.y + (cc * maxSy + 0.5f * res.y)) / res.y);
    rayRes[0] = res.x; rayRes[1] = res.y; rayRes[2] = res.z;
    rayOffset[0] = res.x * (float)lSteps; rayOffset[1] = res.y * (float)bSteps; rayOffset[2] = sitg.offset.z;
    rayDims[0] = (unsigned int)round_to(rSteps - lSteps + 1, SUPERP_TILE_X);
    rayDims[1] = (unsigned int)round_to(tSteps - bSteps + 1, SUPERP_TILE_Y);
    rayDims[2] = b->n_layers;
    return RTD_OK;
}
int orc_ray_grid(const rtd_beam* b, const rtd_options* opt, unsigned int rayDims[3], float rayRes[3], float rayOffset[3]) {
    return ray_grid(b, opt, rayDims, rayRes, rayOffset);
}

int orc_field_run(const rtd_luts* l, const float* ct, const uint32_t ctDims[3], const rtd_beam* b,
                  float* dose, const uint32_t doseDims[3], const rtd_options* opt, int keepLayers, orc_field* out) {
    orc_field f = (orc_field)calloc(1, sizeof(struct orc_field_s));
    if (out) *out = f;
    unsigned int rayDims[3]; float rayRes[3], rayOffset[3];
    int st = ray_grid(b, opt, rayDims, rayRes, rayOffset);
    if (st != RTD_OK) { if (!out) orc_field_free(f); return st; }
    const int W = (int)rayDims[0], H = (int)rayDims[1], L = (int)rayDims[2], S = (int)b->tracer_steps;
    const size_t R = (size_t)W * H;
    f->W = W; f->H = H; f->L = L; f->S = S; f->keepLayers = keepLayers;
    memcpy(f->info.ray_dims, rayDims, sizeof rayDims);
    memcpy(f->info.ray_res, rayRes, sizeof rayRes);
    memcpy(f->info.ray_offset, rayOffset, sizeof rayOffset);

    rtd_idx_transform primRayIdxToGantry;
    for (int i = 0; i < 3; ++i) { primRayIdxToGantry.delta[i] = rayRes[i]; primRayIdxToGantry.offset[i] = rayOffset[i]; }
    fromfan rayIdxToImIdx = make_fromfan(&primRayIdxToGantry, b->source_dist, &b->gantry_to_im_idx);   /* :656-657 */
    const idxtr sitg = idx_from(&b->spot_idx_to_gantry);
    const int ctd[3] = { (int)ctDims[0], (int)ctDims[1], (int)ctDims[2] };

    f->density = (float*)malloc(sizeof(float) * R * S);
    f->wepl = (float*)malloc(sizeof(float) * R * S);
    f->firstInside = (int*)malloc(sizeof(int) * R);
    f->firstOutside = (int*)malloc(sizeof(int) * R);
    f->weplMin = (float*)malloc(sizeof(float) * S);
    f->rayWeights = (float*)malloc(sizeof(float) * R * L);
    f->firstPassive = (int*)calloc(R * L, sizeof(int));
    const size_t layerElems = R * (size_t)S;
    const size_t nKeep = keepLayers ? (size_t)L : 1;
    f->idd = (float*)calloc(layerElems * nKeep, sizeof(float));
    f->rsigma = (float*)calloc(layerElems * nKeep, sizeof(float));
    const int tilesX = W / SUPERP_TILE_X, tilesY = H / SUPERP_TILE_Y;
    const size_t tileElems = (size_t)S * tilesX * tilesY;
    f->tileRad = (uint8_t*)malloc(tileElems * nKeep);
    memset(f->tileRad, 0xFF, tileElems * nKeep);
    f->effRad = (int*)calloc((size_t)L * (MAX_SUPERP_R + 2), sizeof(int));
    f->layerPlan = (float*)calloc((size_t)L * 8, sizeof(float));

    /* K1 tracer (:766-771) */
    tracer_params tp = tracer_params_make(l->density_scale_fact, l->sp_scale_fact, (unsigned int)S, &rayIdxToImIdx);
    stage_trace(ct, ctd, l, &tp, W, H, f->density, f->wepl, f->firstInside, f->firstOutside);

    /* K2 reductions (:781-790) */
    int beamFirstInside = f->firstInside[0], beamFirstOutside = f->firstOutside[0];
    for (size_t i = 1; i < R; ++i) {
        if (f->firstInside[i] < beamFirstInside) beamFirstInside = f->firstInside[i];
        if (f->firstOutside[i] > beamFirstOutside) beamFirstOutside = f->firstOutside[i];
    }
    for (int k = 0; k < S; ++k) {
        const float* p = f->wepl + (size_t)k * R; float m = p[0];
        for (size_t i = 1; i < R; ++i) if (p[i] < m) m = p[i];
        f->weplMin[k] = m;
    }
    const float entryZ = ((float)beamFirstInside) * rayRes[2] + rayOffset[2];                            /* :784 */

    /* cut-offs (:792-802) */
    const float maxEnergy = orc_find_max(b->energies, L);
    const float maxEnergyIdx = orc_find_decimal_ordered(l->energies_per_u, l->n_energies, maxEnergy);
    const float maxPeakDepth = orc_vector_interpolate(l->peak_depths, l->n_energies, maxEnergyIdx);
    const int firstPastCutoffAll = orc_find_first_larger_ordered(f->weplMin, S, opt->bp_depth_cutoff * maxPeakDepth);
    const int beamFirstGuaranteedPassive = firstPastCutoffAll < beamFirstOutside ? firstPastCutoffAll : beamFirstOutside;
    int beamFirstCalculatedPassive = 0;
    f->info.beam_first_inside = beamFirstInside;
    f->info.beam_first_outside = beamFirstOutside;
    f->info.beam_first_guaranteed_passive = beamFirstGuaranteedPassive;

    const int bevW = W + 2 * MAX_SUPERP_R, bevH = H + 2 * MAX_SUPERP_R;
    /* the reference allocates beamFirstGuaranteedPassive slices (:801); we allocate S so tests can compare whole arrays */
    f->bev = (float*)calloc((size_t)bevW * bevH * S, sizeof(float));

    /* per-layer tables (:829-849) */
    float* energyIdcs = (float*)malloc(sizeof(float) * L), *energyScaleFacts = (float*)malloc(sizeof(float) * L);
    float* peakDepths = (float*)malloc(sizeof(float) * L);
    f2* entrySigmas = (f2*)malloc(sizeof(f2) * L);
    for (int ln = 0; ln < L; ++ln) {
        float e = b->energies[ln];
        energyIdcs[ln] = orc_find_decimal_ordered(l->energies_per_u, l->n_energies, e);
        energyScaleFacts[ln] = orc_vector_interpolate(l->scale_facts, l->n_energies, energyIdcs[ln]);
        peakDepths[ln] = orc_vector_interpolate(l->peak_depths, l->n_energies, energyIdcs[ln]);
        f2 c = sigma_sq_air_coefs(peakDepths[ln], opt->nozzle);
        float sx = b->spot_sigmas[2 * ln], sy = b->spot_sigmas[2 * ln + 1];
        entrySigmas[ln].x = sqrtf(c.x * entryZ * entryZ + c.y * entryZ + sx * sx);
        entrySigmas[ln].y = sqrtf(c.x * entryZ * entryZ + c.y * entryZ + sy * sy);
        if (opt->nuclear_corr == RTD_NUC_GAUSS_FIT) { entrySigmas[ln].x = 0.97f * entrySigmas[ln].x; entrySigmas[ln].y = 0.97f * entrySigmas[ln].y; }  /* :842-847 */
    }
    f2 pxSpMult; pxSpMult.x = 1.0f - entryZ / b->source_dist[0]; pxSpMult.y = 1.0f - entryZ / b->source_dist[1];  /* :849 */

    /* K4/K5 spot -> ray weights (:851-854) */
    {
        const unsigned int inDims[3] = { b->spot_nx, b->spot_ny, b->n_layers };
        float* interm = (float*)malloc(sizeof(float) * (size_t)W * b->spot_ny * L);
        stage_conv2d(b->spot_weights, interm, f->rayWeights, entrySigmas, inDims, rayDims, sitg.delta, sitg.offset,
                     mk3(rayRes[0], rayRes[1], rayRes[2]), mk3(rayOffset[0], rayOffset[1], rayOffset[2]), pxSpMult,
                     opt->conv_sigma_cutoff);
        free(interm);
    }

    /* NUCLEAR_CORR set-up (:665-668, 736-751, 858-892): the nuclear grid is the spot grid rounded up to whole tiles */
    const int nucOn = opt->nuclear_corr != RTD_NUC_OFF;
    if (nucOn && (!l->nuc_weight_matrix || !l->nuc_sq_sigma_matrix)) {
        snprintf(f->err, sizeof f->err, "nuclear_corr set but the LUTs carry no nuclear tables");
        free(energyIdcs); free(energyScaleFacts); free(peakDepths); free(entrySigmas);
        if (!out) orc_field_free(f);
        return RTD_ERR_INVALID_ARG;
    }
    const int nucW = nucOn ? round_to((int)b->spot_nx, SUPERP_TILE_X) : 0, nucH = nucOn ? round_to((int)b->spot_ny, SUPERP_TILE_Y) : 0;
    const size_t nucR = (size_t)nucW * nucH, nucIddN = nucR * (size_t)S;
    const int bevNucW = nucW + 2 * MAX_SUPERP_R, bevNucH = nucH + 2 * MAX_SUPERP_R;
    float* nucRayWeights = NULL, *nucIdd = NULL, *nucRs = NULL, *bevNuc = NULL;
    int* nucSpotIdx = NULL;
    uint8_t* nucTileRad = NULL;
    if (nucOn) {
        nucRayWeights = (float*)calloc(nucR * (size_t)L, sizeof(float));                                /* extendAndPadd :51-66 */
        for (int z = 0; z < L; ++z) for (unsigned int y = 0; y < b->spot_ny; ++y) for (unsigned int x = 0; x < b->spot_nx; ++x)
            nucRayWeights[(size_t)z * nucR + (size_t)y * nucW + x] = b->spot_weights[((size_t)z * b->spot_ny + y) * b->spot_nx + x];
        nucIdd = (float*)calloc(nucIddN, sizeof(float));                                                /* :862 */
        nucRs = (float*)malloc(sizeof(float) * nucIddN);                                                /* :863 */
        for (size_t i = 0; i < nucIddN; ++i) nucRs[i] = INFINITY;
        bevNuc = (float*)calloc((size_t)bevNucW * bevNucH * S, sizeof(float));
        nucTileRad = (uint8_t*)malloc((size_t)S * (nucW / SUPERP_TILE_X) * (nucH / SUPERP_TILE_Y));
        nucSpotIdx = (int*)malloc(sizeof(int) * R);                                                     /* :878-892 */
        for (size_t i = 0; i < R; ++i) nucSpotIdx[i] = -1;
        for (unsigned int sy = 0; sy < b->spot_ny; ++sy) {
            float gantryPosY = (float)sy * sitg.delta.y + sitg.offset.y;
            int rayIdxY = (int)roundf((gantryPosY - rayOffset[1]) / rayRes[1]);
            for (unsigned int sx = 0; sx < b->spot_nx; ++sx) {
                float gantryPosX = (float)sx * sitg.delta.x + sitg.offset.x;
                int rayIdxX = (int)roundf((gantryPosX - rayOffset[0]) / rayRes[0]);
                if (rayIdxX >= 0 && rayIdxX < W && rayIdxY >= 0 && rayIdxY < H)      /* (the reference indexes unchecked) */
                    nucSpotIdx[(size_t)W * rayIdxY + rayIdxX] = nucW * (int)sy + (int)sx;
            }
        }
    }

    int status = RTD_OK;
    int maxRadius = 0;
    int64_t liveSteps = 0;
    if (beamFirstGuaranteedPassive > beamFirstInside) {
        for (int ln = 0; ln < L; ++ln) {                                                               /* :916 */
            unsigned int localAfterLast = (unsigned int)orc_find_first_larger_ordered(f->weplMin, S, opt->bp_depth_cutoff * peakDepths[ln]);
            unsigned int afterLastStep = localAfterLast < (unsigned int)beamFirstGuaranteedPassive ? localAfterLast : (unsigned int)beamFirstGuaranteedPassive;
            fill_params fp = fill_params_make(energyIdcs[ln], energyScaleFacts[ln], peakDepths[ln], l->rrl_scale_fact,
                                              (unsigned int)beamFirstInside, afterLastStep, &rayIdxToImIdx, opt->nozzle);
            float* idd = f->idd + (keepLayers ? layerElems * ln : 0);
            float* rs = f->rsigma + (keepLayers ? layerElems * ln : 0);
            uint8_t* tr = f->tileRad + (keepLayers ? tileElems * ln : 0);
            int* fpass = f->firstPassive + R * ln;
            nuc_fill nf;
            if (nucOn) {
                nf.bevNucIdd = nucIdd; nf.bevNucRSigmaEff = nucRs; nf.nucRayWeights = nucRayWeights + nucR * ln; nf.nucIdcs = nucSpotIdx;
                nf.spotDist = sitg.delta.x / b->ray_spacing[0];                                          /* spotDistInRays :922 */
                nf.entrySigmaSq = entrySigmas[ln].x * entrySigmas[ln].x;                                  /* :925, 4th argument */
                nf.nucMemStep = 0;                                                                       /* :925, 7th argument */
            }
            stage_fill(f->density, f->wepl, idd, rs, f->rayWeights + R * ln, f->firstInside, f->firstOutside, fpass,
                       &fp, l, opt, W, H, nucOn ? &nf : NULL);
            int layerFirstPassive = fpass[0];                                                          /* :952-957 */
            for (size_t i = 1; i < R; ++i) if (fpass[i] > layerFirstPassive) layerFirstPassive = fpass[i];
            if (layerFirstPassive > beamFirstCalculatedPassive) beamFirstCalculatedPassive = layerFirstPassive;
            float* lp = f->layerPlan + (size_t)ln * 8;
            lp[0] = energyIdcs[ln]; lp[1] = energyScaleFacts[ln]; lp[2] = peakDepths[ln];
            lp[3] = entrySigmas[ln].x; lp[4] = entrySigmas[ln].y; lp[5] = (float)afterLastStep; lp[6] = (float)layerFirstPassive;

            int ctrs[MAX_SUPERP_R + 2];
            stage_tile_radius(rs, W, H, beamFirstInside, layerFirstPassive, opt->ks_sigma_cutoff, tr, S, ctrs);  /* :959-963 */
            int* eff = f->effRad + (size_t)ln * (MAX_SUPERP_R + 2);
            if (ctrs[MAX_SUPERP_R + 1] > 0) {                                                          /* :965 */
                snprintf(f->err, sizeof f->err, "Found larger than allowed kernel superposition radius");
                status = RTD_ERR_RADIUS_OVERFLOW;
                break;
            }
            int layerMax = orc_batch_radii(ctrs, eff);
            if (layerMax > maxRadius) maxRadius = layerMax;
            if (layerFirstPassive > beamFirstInside) liveSteps += layerFirstPassive - beamFirstInside;
            stage_superposition(idd, rs, f->bev, W, H, tr, eff, beamFirstInside, layerFirstPassive);    /* :1024-1056 */
            if (nucOn) {                                                                                /* :978-997, :1058-1091 */
                int nctrs[MAX_SUPERP_R + 2], neff[MAX_SUPERP_R + 2];
                stage_tile_radius(nucRs, nucW, nucH, beamFirstInside, layerFirstPassive, opt->ks_sigma_cutoff, nucTileRad, S, nctrs);
                if (nctrs[MAX_SUPERP_R + 1] > 0) {
                    snprintf(f->err, sizeof f->err, "Found larger than allowed kernel superposition radius");
                    status = RTD_ERR_RADIUS_OVERFLOW;
                    break;
                }
                orc_batch_radii(nctrs, neff);
                stage_superposition(nucIdd, nucRs, bevNuc, nucW, nucH, nucTileRad, neff, beamFirstInside, layerFirstPassive);
            }
        }
    }
    f->info.beam_first_calculated_passive = beamFirstCalculatedPassive;
    f->info.live_steps = liveSteps;
    f->info.max_radius = maxRadius;

    /* transfer (:1107-1218) */
    if (status == RTD_OK && beamFirstCalculatedPassive > beamFirstInside) {
        fromfan primRayIdxToDoseIdx = make_fromfan(&primRayIdxToGantry, b->source_dist, &b->gantry_to_dose_idx);  /* :1185 */
        f3 maxP = mk3(-1.0f, -1.0f, -1.0f), minP = mk3(100000.0f, 100000.0f, 100000.0f);
        float xVals[2] = { -(float)MAX_SUPERP_R, (float)(W + MAX_SUPERP_R - 1) };
        float yVals[2] = { -(float)MAX_SUPERP_R, (float)(H + MAX_SUPERP_R - 1) };
        float zVals[2] = { (float)beamFirstInside, (float)(beamFirstCalculatedPassive - 1) };
        for (int zi = 0; zi < 2; ++zi) for (int yi = 0; yi < 2; ++yi) for (int xi = 0; xi < 2; ++xi) {
            f3 p = fromfan_point(&primRayIdxToDoseIdx, mk3(xVals[xi], yVals[yi], zVals[zi]));
            if (p.x > maxP.x) maxP.x = p.x; if (p.y > maxP.y) maxP.y = p.y; if (p.z > maxP.z) maxP.z = p.z;
            if (p.x < minP.x) minP.x = p.x; if (p.y < minP.y) minP.y = p.y; if (p.z < minP.z) minP.z = p.z;
        }
        int minIdx[3], maxIdx[3];
        int t;
        t = (((int)floorf(minP.x)) / 32) * 32; minIdx[0] = t > 0 ? t : 0;                                /* :1207 */
        t = (int)floorf(minP.y); minIdx[1] = t > 0 ? t : 0;
        t = (int)floorf(minP.z); minIdx[2] = t > 0 ? t : 0;
        t = (int)ceilf(maxP.x); maxIdx[0] = t < (int)doseDims[0] - 1 ? t : (int)doseDims[0] - 1;         /* :1208 */
        t = (int)ceilf(maxP.y); maxIdx[1] = t < (int)doseDims[1] - 1 ? t : (int)doseDims[1] - 1;
        t = (int)ceilf(maxP.z); maxIdx[2] = t < (int)doseDims[2] - 1 ? t : (int)doseDims[2] - 1;
        memcpy(f->info.bbox_min, minIdx, sizeof minIdx);
        memcpy(f->info.bbox_max, maxIdx, sizeof maxIdx);
        tofan inv = fromfan_invert_and_shift(&primRayIdxToDoseIdx, mk3((float)MAX_SUPERP_R, (float)MAX_SUPERP_R, -(float)beamFirstInside)); /* :1213 */
        transfer_params tps = transfer_params_make(&inv);
        if (maxIdx[0] >= minIdx[0] && maxIdx[1] >= minIdx[1] && maxIdx[2] >= minIdx[2])
            stage_transfer(dose, doseDims, &tps, minIdx, maxIdx, f->bev + (size_t)beamFirstInside * bevW * bevH, bevW, bevH,
                           beamFirstCalculatedPassive - beamFirstInside);
    }
    /* nuclear transfer (:1221-1254): the halo cube lives on the spot grid -> its own fan transform */
    if (nucOn && status == RTD_OK && beamFirstCalculatedPassive > beamFirstInside) {
        fromfan nucRayIdxToDoseIdx = make_fromfan(&b->spot_idx_to_gantry, b->source_dist, &b->gantry_to_dose_idx);
        f3 maxP = mk3(-1.0f, -1.0f, -1.0f), minP = mk3(100000.0f, 100000.0f, 100000.0f);
        float xVals[2] = { -(float)MAX_SUPERP_R, (float)(nucW + MAX_SUPERP_R - 1) };
        float yVals[2] = { -(float)MAX_SUPERP_R, (float)(nucH + MAX_SUPERP_R - 1) };
        float zVals[2] = { (float)beamFirstInside, (float)(beamFirstCalculatedPassive - 1) };
        for (int zi = 0; zi < 2; ++zi) for (int yi = 0; yi < 2; ++yi) for (int xi = 0; xi < 2; ++xi) {
            f3 p = fromfan_point(&nucRayIdxToDoseIdx, mk3(xVals[xi], yVals[yi], zVals[zi]));
            if (p.x > maxP.x) maxP.x = p.x; if (p.y > maxP.y) maxP.y = p.y; if (p.z > maxP.z) maxP.z = p.z;
            if (p.x < minP.x) minP.x = p.x; if (p.y < minP.y) minP.y = p.y; if (p.z < minP.z) minP.z = p.z;
        }
        int minIdx[3], maxIdx[3], t;
        t = (((int)floorf(minP.x)) / 32) * 32; minIdx[0] = t > 0 ? t : 0;
        t = (int)floorf(minP.y); minIdx[1] = t > 0 ? t : 0;
        t = (int)floorf(minP.z); minIdx[2] = t > 0 ? t : 0;
        t = (int)ceilf(maxP.x); maxIdx[0] = t < (int)doseDims[0] - 1 ? t : (int)doseDims[0] - 1;
        t = (int)ceilf(maxP.y); maxIdx[1] = t < (int)doseDims[1] - 1 ? t : (int)doseDims[1] - 1;
        t = (int)ceilf(maxP.z); maxIdx[2] = t < (int)doseDims[2] - 1 ? t : (int)doseDims[2] - 1;
        tofan inv = fromfan_invert_and_shift(&nucRayIdxToDoseIdx, mk3((float)MAX_SUPERP_R, (float)MAX_SUPERP_R, -(float)beamFirstInside));
        transfer_params tps = transfer_params_make(&inv);
        if (maxIdx[0] >= minIdx[0] && maxIdx[1] >= minIdx[1] && maxIdx[2] >= minIdx[2])
            stage_transfer(dose, doseDims, &tps, minIdx, maxIdx, bevNuc + (size_t)beamFirstInside * bevNucW * bevNucH, bevNucW, bevNucH,
                           beamFirstCalculatedPassive - beamFirstInside);
    }
    free(nucRayWeights); free(nucIdd); free(nucRs); free(bevNuc); free(nucSpotIdx); free(nucTileRad);
    free(energyIdcs); free(energyScaleFacts); free(peakDepths); free(entrySigmas);
    if (!out) orc_field_free(f);
    return status;
}

/* The reference-shaped call: accumulate all beams into dose (kernel_wrapper.cu:381-1369). */
int orc_compute(const rtd_luts* l, const float* ct, const uint32_t ctDims[3], const rtd_beam* beams, int nBeams,
                float* dose, const uint32_t doseDims[3], const rtd_options* opt) {
    for (int i = 0; i < nBeams; ++i) {
        int st = orc_field_run(l, ct, ctDims, &beams[i], dose, doseDims, opt, 0, NULL);
        if (st != RTD_OK) return st;
    }
    return RTD_OK;
}

const char* orc_field_error(orc_field f) { return f ? f->err : ""; }
void orc_field_info(orc_field f, rtd_field_info* out) { *out = f->info; }

/* Same names as rtd_field_fetch (include/rtd.h). Returns a pointer into the field and the byte count. */
const void* orc_field_get(orc_field f, const char* name, size_t* bytes) {
    const size_t R = (size_t)f->W * f->H, S = (size_t)f->S, L = (size_t)f->L;
    const size_t nKeep = f->keepLayers ? L : 1;
    const size_t tiles = (size_t)(f->W / SUPERP_TILE_X) * (f->H / SUPERP_TILE_Y);
    if (!strcmp(name, "density")) { *bytes = 4 * R * S; return f->density; }
    if (!strcmp(name, "wepl")) { *bytes = 4 * R * S; return f->wepl; }
    if (!strcmp(name, "first_inside")) { *bytes = 4 * R; return f->firstInside; }
    if (!strcmp(name, "first_outside")) { *bytes = 4 * R; return f->firstOutside; }
    if (!strcmp(name, "wepl_min")) { *bytes = 4 * S; return f->weplMin; }
    if (!strcmp(name, "ray_weights")) { *bytes = 4 * R * L; return f->rayWeights; }
    if (!strcmp(name, "idd")) { *bytes = 4 * R * S * nKeep; return f->idd; }
    if (!strcmp(name, "rsigma")) { *bytes = 4 * R * S * nKeep; return f->rsigma; }
    if (!strcmp(name, "first_passive")) { *bytes = 4 * R * L; return f->firstPassive; }
    if (!strcmp(name, "tile_radius")) { *bytes = S * tiles * nKeep; return f->tileRad; }
    if (!strcmp(name, "eff_radius")) { *bytes = 4 * L * (MAX_SUPERP_R + 2); return f->effRad; }
    if (!strcmp(name, "bev")) { *bytes = 4 * (size_t)(f->W + 64) * (f->H + 64) * S; return f->bev; }
    if (!strcmp(name, "layer_plan")) { *bytes = 4 * L * 8; return f->layerPlan; }
    *bytes = 0; return NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* Separable uniform-sigma convolution, the algorithm of the reference's (unbuilt) CPU files    */
/* cpu_convolution_1d.cpp:36-61 (xConvCpu, gather) and :145-171 (yConvCpu, scatter). Pinned      */
/* bit-for-bit against the compiled reference in tests/test_oracle_golden.py.                   */

static void erf_diffs_cpu(float rSigmaEff, unsigned int rad, float* e) { /* cpu_convolution_1d.cpp:38-46 */
    float erfNew = erff(rSigmaEff * 0.5f), erfOld;
    e[0] = erfNew;
    for (unsigned int i = 1; i < rad + 1; ++i) {
        erfOld = erfNew;
        erfNew = erff(rSigmaEff * ((float)i + 0.5f));
        e[i] = 0.5f * (erfNew - erfOld);
    }
}
void orc_x_conv_cpu(const float* in, float* out, float rSigmaEff, unsigned int rad, unsigned int inWidth,
                    unsigned int outWidth, unsigned int height, int inOutOffset) {
    float* e = (float*)malloc(sizeof(float) * (rad + 1));
    erf_diffs_cpu(rSigmaEff, rad, e);
    for (unsigned int y = 0; y < height; ++y) {
        for (unsigned int xOut = 0; xOut < outWidth; ++xOut) {
            float res = 0.0f;
            for (long i = -(long)rad; i < (long)rad + 1; ++i) {
                /* the reference computes xOut - inOutOffset in UNSIGNED arithmetic (cpu_convolution_1d.cpp:53), so
                 * outputs left of inOutOffset wrap out of range and stay 0: reproduced, it is what the code does */
                long xIn = (long)(unsigned int)(xOut - (unsigned int)inOutOffset) + i;
                if (xIn >= 0 && xIn < (long)inWidth) res += e[labs(i)] * in[(size_t)y * inWidth + xIn];
            }
            out[(size_t)y * outWidth + xOut] = res;
        }
    }
    free(e);
}
/* xConvCpuScat (cpu_convolution_1d.cpp:63-89): the scatter form, accumulates into out (which the caller zeroes); unlike the
 * gather above it fills the left apron too. rad <= inOutOffset is the reference's precondition (it exits otherwise). */
void orc_x_conv_cpu_scat(const float* in, float* out, float rSigmaEff, unsigned int rad, unsigned int inWidth,
                         unsigned int outWidth, unsigned int height, unsigned int inOutOffset) {
    float* e = (float*)malloc(sizeof(float) * (rad + 1));
    erf_diffs_cpu(rSigmaEff, rad, e);
    for (unsigned int y = 0; y < height; ++y) {
        for (unsigned int xIn = 0; xIn < inWidth; ++xIn) {
            const float val = in[(size_t)y * inWidth + xIn];
            for (long i = -(long)rad; i < (long)rad + 1; ++i) {
                long xOut = (long)xIn + (long)inOutOffset + i;
                out[(size_t)y * outWidth + xOut] += e[labs(i)] * val;
            }
        }
    }
    free(e);
}
void orc_y_conv_cpu(const float* in, float* out, float rSigmaEff, unsigned int rad, unsigned int inHeight,
                    unsigned int width, int inOutOffset) {
    float* e = (float*)malloc(sizeof(float) * (rad + 1));
    erf_diffs_cpu(rSigmaEff, rad, e);
    for (unsigned int yIn = 0; yIn < inHeight; ++yIn) {
        for (long i = -(long)rad; i < (long)rad + 1; ++i) {
            long yOut = (long)yIn + inOutOffset + i;
            float w = e[labs(i)];
            for (unsigned int x = 0; x < width; ++x) out[(size_t)yOut * width + x] += w * in[(size_t)yIn * width + x];
        }
    }
    free(e);
}

/* ------------------------------------------------------------------------------------------ */
/* LUT text reader: restatement of energy_reader.cpp:12-101 (whitespace separated ASCII).       */

static int read_floats(FILE* fp, float* dst, size_t n) {
    for (size_t i = 0; i < n; ++i) if (fscanf(fp, "%f", &dst[i]) != 1) return -1;
    return 0;
}
static int read_1d(const char* dir, const char* name, int32_t* n, float* scale, float** vec) {
    char path[4096];
    snprintf(path, sizeof path, "%s%s", dir, name);
    FILE* fp = fopen(path, "r");
    if (!fp) return RTD_ERR_IO;
    if (fscanf(fp, "%d %f", n, scale) != 2 || *n <= 0) { fclose(fp); return RTD_ERR_IO; }
    *vec = (float*)malloc(sizeof(float) * (size_t)*n);
    int rc = read_floats(fp, *vec, (size_t)*n);
    fclose(fp);
    return rc ? RTD_ERR_IO : RTD_OK;
}
/* dir must end with '/' exactly like PHYS_DATA_DIRECTORY (CMakeLists.txt:32). Arrays are malloc'd; free with orc_luts_free. */
int orc_read_luts(const char* dir, int waterCubeTest, rtd_luts* out) {
    memset(out, 0, sizeof *out);
    char path[4096];
    snprintf(path, sizeof path, "%sproton_cumul_ddd_data.txt", dir);
    FILE* fp = fopen(path, "r");
    if (!fp) return RTD_ERR_IO;
    if (fscanf(fp, "%d %d", &out->n_energy_samples, &out->n_energies) != 2) { fclose(fp); return RTD_ERR_IO; }
    size_t nE = (size_t)out->n_energies, nS = (size_t)out->n_energy_samples;
    float* e = (float*)malloc(sizeof(float) * nE), *p = (float*)malloc(sizeof(float) * nE), *s = (float*)malloc(sizeof(float) * nE);
    float* m = (float*)malloc(sizeof(float) * nE * nS);
    int rc = read_floats(fp, e, nE) | read_floats(fp, p, nE) | read_floats(fp, s, nE) | read_floats(fp, m, nE * nS);
    fclose(fp);
    out->energies_per_u = e; out->peak_depths = p; out->scale_facts = s; out->cidd_matrix = m;
    if (rc) return RTD_ERR_IO;
    float* v = NULL;
    if ((rc = read_1d(dir, "density_Schneider2000_adj.txt", &out->n_density_samples, &out->density_scale_fact, &v)) != RTD_OK) return rc;
    out->density_vector = v; v = NULL;
    if ((rc = read_1d(dir, "HU_to_SP_H&N_adj.txt", &out->n_sp_samples, &out->sp_scale_fact, &v)) != RTD_OK) return rc;
    out->sp_vector = v; v = NULL;
    if ((rc = read_1d(dir, waterCubeTest ? "radiation_length_inc_water.txt" : "radiation_length.txt",
                      &out->n_rrl_samples, &out->rrl_scale_fact, &v)) != RTD_OK) return rc;
    out->rrl_vector = v;
    return RTD_OK;
}
/* The NUCLEAR_CORR part of energyReader (energy_reader.cpp:103-162): the variant's table, with the reference's consistency
 * checks against the cumulative-IDD table. variant: RTD_NUC_*. */
int orc_read_luts_nuc(const char* dir, int waterCubeTest, int variant, rtd_luts* out) {
    int rc = orc_read_luts(dir, waterCubeTest, out);
    if (rc != RTD_OK || variant == RTD_NUC_OFF) return rc;
    const char* name = variant == RTD_NUC_SOUKUP ? "nuclear_weights_and_sigmas_Soukup.txt"
                     : variant == RTD_NUC_FLUKA ? "nuclear_weights_and_sigmas_Fluka.txt" : "nuclear_weights_and_sigmas_fit.txt";
    char path[4096];
    snprintf(path, sizeof path, "%s%s", dir, name);
    FILE* fp = fopen(path, "r");
    if (!fp) return RTD_ERR_IO;
    int nS = 0, nE = 0;
    if (fscanf(fp, "%d %d", &nS, &nE) != 2 || nS != out->n_energy_samples || nE != out->n_energies) { fclose(fp); return RTD_ERR_IO; }
    const float* axes[3] = { out->energies_per_u, out->peak_depths, out->scale_facts };
    for (int a = 0; a < 3; ++a)
        for (int i = 0; i < nE; ++i) {
            float v;
            if (fscanf(fp, "%f", &v) != 1 || fabsf(axes[a][i] - v) > 0.01f) { fclose(fp); return RTD_ERR_IO; }
        }
    float* w = (float*)malloc(sizeof(float) * (size_t)nE * nS), *q = (float*)malloc(sizeof(float) * (size_t)nE * nS);
    rc = read_floats(fp, w, (size_t)nE * nS) | read_floats(fp, q, (size_t)nE * nS);
    fclose(fp);
    out->nuc_weight_matrix = w; out->nuc_sq_sigma_matrix = q;
    return rc ? RTD_ERR_IO : RTD_OK;
}
void orc_luts_free(rtd_luts* l) {
    free((void*)l->energies_per_u); free((void*)l->peak_depths); free((void*)l->scale_facts); free((void*)l->cidd_matrix);
    free((void*)l->density_vector); free((void*)l->sp_vector); free((void*)l->rrl_vector);
    free((void*)l->nuc_weight_matrix); free((void*)l->nuc_sq_sigma_matrix);
    memset(l, 0, sizeof *l);
}

/* ------------------------------------------------------------------------------------------ */
/* Gamma index (global, dose difference dd of max(ref), distance dta mm) for the parity report. */
/* Not part of the reference (it has no evaluator); kept here with the checker.     */
/* Brute-force search over evaluated voxels within a cube of +-ceil(dta*searchMult/spacing),    */
/* evaluated only where ref >= threshold*max(ref). Returns pass fraction; *nEval = voxels used. */
double orc_gamma_pass_rate(const float* ref, const float* eval, const uint32_t dims[3], const float spacing[3],
                           float ddFrac, float dtaMm, float thresholdFrac, int64_t* nEval, float* maxGamma) {
    const int nx = (int)dims[0], ny = (int)dims[1], nz = (int)dims[2];
    const size_t n = (size_t)nx * ny * nz;
    float maxRef = 0.0f;
    for (size_t i = 0; i < n; ++i) if (ref[i] > maxRef) maxRef = ref[i];
    if (!(maxRef > 0.0f)) { *nEval = 0; if (maxGamma) *maxGamma = 0.0f; return 1.0; }
    const float dd = ddFrac * maxRef, thr = thresholdFrac * maxRef;
    const int rx = (int)ceilf(1.5f * dtaMm / spacing[0]), ry = (int)ceilf(1.5f * dtaMm / spacing[1]), rz = (int)ceilf(1.5f * dtaMm / spacing[2]);
    int64_t cnt = 0, pass = 0; float gmax = 0.0f;
#pragma omp parallel for reduction(+:cnt, pass) reduction(max:gmax) schedule(dynamic, 1)
    for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
        const float r = ref[((size_t)z * ny + y) * nx + x];
        if (r < thr) continue;
        ++cnt;
        float best = INFINITY;
        for (int dz = -rz; dz <= rz; ++dz) { int zz = z + dz; if (zz < 0 || zz >= nz) continue;
        for (int dy = -ry; dy <= ry; ++dy) { int yy = y + dy; if (yy < 0 || yy >= ny) continue;
        for (int dx = -rx; dx <= rx; ++dx) { int xx = x + dx; if (xx < 0 || xx >= nx) continue;
            float dist2 = (dx * spacing[0]) * (dx * spacing[0]) + (dy * spacing[1]) * (dy * spacing[1]) + (dz * spacing[2]) * (dz * spacing[2]);
            float dv = eval[((size_t)zz * ny + yy) * nx + xx] - r;
            float g2 = dist2 / (dtaMm * dtaMm) + (dv * dv) / (dd * dd);
            if (g2 < best) best = g2;
        }}}
        float g = sqrtf(best);
        if (g <= 1.0f) ++pass;
        if (g > gmax) gmax = g;
    }
    *nEval = cnt;
    if (maxGamma) *maxGamma = gmax;
    return cnt ? (double)pass / (double)cnt : 1.0;
}
