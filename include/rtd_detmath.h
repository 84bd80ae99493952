/*
 * rtd_detmath.h — a power function and an error function that give the SAME BITS on the host CPU and on the GPU.
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
 *
 * rtd_erf_det stands in for the device erff of the spot -> ray convolution (src/gpu_convolution_2d.cu:27,51), whose
 * result is compared with RAY_WEIGHT_CUTOFF ray by ray (src/kernel_wrapper.cu:244): with one error function on both
 * machines the ray weights, and with them the set of live rays, are the same bits. |absolute error| <= 1.0e-7 (about
 * 1.5 ulp at erf ~ 0.8), relative error <= 8e-8 for |x| < 0.875; coefficients derived by tools/fit_detmath.py.
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

/* erf(a): a + a*R(a^2) for |a| < 0.875; 1 - 2^P(|a| - 0.875) up to 4; 1 beyond (erfc(4) < 2^-25) */
RTD_DM_FN float rtd_erf_det(float a) {
    const float t = fabsf(a);
    if (t < 0.875f) {
        const float s = a * a;
        float r = 8.694667811e-05f;
        r = fmaf(r, s, -8.215559851e-04f);
        r = fmaf(r, s, 5.207134257e-03f);
        r = fmaf(r, s, -2.686173980e-02f);
        r = fmaf(r, s, 1.128373582e-01f);
        r = fmaf(r, s, -3.761263625e-01f);
        r = fmaf(r, s, 1.283791669e-01f);
        return fmaf(r, a, a);
    }
    if (!(t < 4.0f)) return copysignf(1.0f, a);
    const float u = t - 0.875f;
    /* log2(erfc(t)), degree 9 in u on [0, 3.125] */
    float p = -3.116289875e-08f;
    p = fmaf(p, u, 2.754377229e-06f);
    p = fmaf(p, u, -5.187475768e-05f);
    p = fmaf(p, u, 5.094422115e-04f);
    p = fmaf(p, u, -3.331390714e-03f);
    p = fmaf(p, u, 1.650326662e-02f);
    p = fmaf(p, u, -6.772692889e-02f);
    p = fmaf(p, u, -1.192432172e+00f);
    p = fmaf(p, u, -3.506067286e+00f);
    p = fmaf(p, u, -2.211398194e+00f);
    const float n = rintf(p);
    const float r = p - n;                         /* exact */
    float q = 1.340043265e-03f;                    /* 2^r as in rtd_pow_det */
    q = fmaf(q, r, 9.676037356e-03f);
    q = fmaf(q, r, 5.550327152e-02f);
    q = fmaf(q, r, 2.402210683e-01f);
    q = fmaf(q, r, 6.931471825e-01f);
    q = fmaf(q, r, 1.000000119e+00f);
    return copysignf(1.0f - ldexpf(q, (int)n), a);
}

#endif /* RTD_DETMATH_H */
