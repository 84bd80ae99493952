"""Derives the polynomial coefficients of include/rtd_detmath.h (rtd_erf_det) and checks the float32 evaluation
against scipy in double precision. Run: python tools/fit_detmath.py"""
import numpy as np
from scipy import special
from numpy.polynomial import chebyshev as C, polynomial as P

X0, X1 = 0.875, 4.0


def cheb_fit(fun, a, b, deg, n=6000):
    k = np.arange(n); x = np.cos(np.pi * (k + 0.5) / n); t = 0.5 * (b - a) * x + 0.5 * (a + b)
    c = C.chebfit(x, fun(t), deg)
    px = P.Polynomial(C.cheb2poly(c)); lin = P.Polynomial([-(a + b) / (b - a), 2 / (b - a)])
    return px(lin).coef


def r1(s):
    x = np.sqrt(np.maximum(s, 1e-300))
    return np.where(s < 1e-20, 2 / np.sqrt(np.pi), special.erf(x) / x) - 1.0


def p2(u):                                   # log2(erfc(t)), t = u + X0
    t = u + X0
    return (special.log_ndtr(-t * np.sqrt(2)) + np.log(2.0)) / np.log(2.0)


c1 = cheb_fit(r1, 0.0, X0 * X0, 6)
c2 = cheb_fit(p2, 0.0, X1 - X0, 9)
f32 = np.float32


def horner32(co, x):
    r = np.full_like(x, f32(co[-1]))
    for c in co[-2::-1]:
        r = (r.astype(np.float64) * x.astype(np.float64) + np.float64(f32(c))).astype(f32)   # fma: one rounding
    return r


EXP2 = [1.000000119e+00, 6.931471825e-01, 2.402210683e-01, 5.550327152e-02, 9.676037356e-03, 1.340043265e-03]


def erf_det32(a):
    a = a.astype(f32); t = np.abs(a)
    s = (a * a).astype(f32)
    small = horner32(c1, s)
    small = (small.astype(np.float64) * a.astype(np.float64) + a.astype(np.float64)).astype(f32)
    u = (t - f32(X0)).astype(f32)
    p = horner32(c2, np.minimum(u, f32(X1 - X0)))
    n = np.rint(p).astype(f32); r = (p - n).astype(f32)
    q = horner32(EXP2, r)
    big = (f32(1.0) - np.ldexp(q, n.astype(np.int32)).astype(f32)).astype(f32)
    big = np.where(t >= f32(X1), f32(1.0), big)
    return np.where(t < f32(X0), small, np.copysign(big, a)).astype(f32)


if __name__ == "__main__":
    x = np.linspace(-5, 5, 4000001).astype(f32)
    e = erf_det32(x).astype(np.float64); ref = special.erf(x.astype(np.float64))
    err = np.abs(e - ref)
    print("max abs err", err.max(), "at", x[err.argmax()])
    m = np.abs(x) < X0
    print("max rel err, small branch", (err[m] / np.maximum(np.abs(ref[m]), 1e-30)).max())
    print("/* erf(x)/x - 1 in s = x^2, |x| < %.3f */" % X0)
    for c in c1[::-1]: print("    %.9ef" % c)
    print("/* log2(erfc(t)) in u = t - %.3f" % X0 + ", %.3f <= t < %.1f */" % (X0, X1))
    for c in c2[::-1]: print("    %.9ef" % c)
