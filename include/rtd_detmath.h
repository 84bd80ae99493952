/*
 * rtd_detmath.h — a power function that gives the SAME BITS on the host CPU and on the GPU.
 *
 * The reference evaluates the residual-range -> energy relation of the sigma recurrence with CUDA's __powf
 * (src/kernel_wrapper.cu:282), i.e. exp2(y * log2 x) on the special-function unit with a few ulp of error that
 * no other machine reproduces bit for bit. Everything downstream of it is index work: the sum of scatter terms
 * becomes 1/sigma, whose tile minimum is thresholded into the superposition radius class
 * (tileRadCalc, src/kernel_wrapper.cuh:300-305), and the batching of radii follows from the class histogram
 * (src/kernel_wrapper.cu:959-976). So that those integers can be compared EXACTLY between the HIP engine and a
 * CPU restatement of the algorithm (the checker of the test suite), both evaluate the power with the routine below: frexp / rint / ldexp (exact operations), explicit
 * fused multiply-adds and nothing else — every step is a correctly rounded IEEE-754 binary32 operation, hence
 * identical on gfx950 (v_frexp_*, v_rndne_f32, v_ldexp_f32, v_fma_f32) and on any host libm.
 *
 * Accuracy: |relative error| <= 2.7e-7 (about 2 ulp) over x in [1e-6, 400], y = 0.5649718 (checked against a
 * double-precision pow by the test suite) — tighter than the intrinsic it stands in for.
 *
 * Domain: x finite, normal and > 0; |y * log2 x| < 126.
 * Include with RTD_DM_FN defined to the function qualifiers wanted (e.g. `__host__ __device__ inline`).
 */
#ifndef RTD_DETMATH_H
#define RTD_DETMATH_H

#include <math.h>

#ifndef RTD_DM_FN
#define RTD_DM_FN static inline
#endif

/* x^y = 2^(y*e) * 2^(y*log2 m),  x = 2^e * m,  m in [sqrt(1/2), sqrt(2)) */
RTD_DM_FN float rtd_pow_det(float x, float y) {
    int e;
    float m = frexpf(x, &e);                       /* m in [0.5, 1), exact */
    if (m < 0.70710678f) { m = m + m; e -= 1; }    /* exact */
    const float f = m - 1.0f;                      /* exact (m within a factor 2 of 1) */
    /* log2(1 + f) = f * P(f) on [-0.2929, 0.4142]; degree 7, |error of f*P| < 1.2e-7 */
    float p = -1.462049037e-01f;
    p = fmaf(p, f, 2.342116535e-01f);
    p = fmaf(p, f, -2.488220930e-01f);
    p = fmaf(p, f, 2.870754302e-01f);
    p = fmaf(p, f, -3.602419496e-01f);
    p = fmaf(p, f, 4.809240401e-01f);
    p = fmaf(p, f, -7.213527560e-01f);
    p = fmaf(p, f, 1.442694902e+00f);
    const float lg = f * p;
    const float ef = (float)e;
    /* n = nearest integer of y*log2 x (any neighbour would do: it only has to be the same on both machines);
     * r = y*e - n + y*lg with the cancellation carried out inside two FMAs, |r| <~ 0.5 */
    const float n = rintf(fmaf(y, lg, y * ef));
    float r = fmaf(y, ef, -n);
    r = fmaf(y, lg, r);
    /* 2^r on [-0.5, 0.5]; degree 5, |error| < 8e-8 */
    float q = 1.340043265e-03f;
    q = fmaf(q, r, 9.676037356e-03f);
    q = fmaf(q, r, 5.550327152e-02f);
    q = fmaf(q, r, 2.402210683e-01f);
    q = fmaf(q, r, 6.931471825e-01f);
    q = fmaf(q, r, 1.000000119e+00f);
    return ldexpf(q, (int)n);                      /* exact scaling */
}

#endif /* RTD_DETMATH_H */
