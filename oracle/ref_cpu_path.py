"""The reference's CPU path (src/cpu_convolution_1d.cpp: xConvCpuScat / yConvCpu) applied to a water field. TEST INFRASTRUCTURE ONLY.

In water every ray of a BEV slice has the same 1/sigma, so the per-voxel-sigma superposition (kernelSuperposition,
kernel_wrapper.cuh:432-489) is a separable convolution of the slice: an x pass and a y pass with the slice's erf-difference
weights — what the reference's CPU convolution routines compute (the scatter form of the x pass: the gather form, xConvCpu, leaves
the left apron at zero through its unsigned index arithmetic, :53). `separable_bev` rebuilds the BEV dose of an oracle field
that way, with either the oracle's restatement of those routines or the reference's own, compiled from its sources where they
lie into oracle/_ref/libref.so (oracle/Makefile). Used by tests/test_oracle_ref_cpu.py (pins the oracle's superposition stage
against the reference's code on BASELINE.json configs[0]) and by bench.py's cpu_baseline leg (times it: kind "reference")."""
import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = None


def ref_lib():
    """oracle/_ref/libref.so, or None when it has not been built (no /root/reference at build time)."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
    return _REF


def _conv_pair(which):
    if which == "reference":
        L = ref_lib()
        if L is None:
            return None
        return L.ref_x_conv_cpu_scat, L.ref_y_conv_cpu
    from oracle import oracle
    L = oracle.lib()
    return L.orc_x_conv_cpu_scat, L.orc_y_conv_cpu


def separable_bev(of, beam, which="reference", repeat=1):
    """(bev [S][H+64][W+64], seconds inside the convolution routines, slices, max radius) of oracle field `of` (run with
    keep_layers=True), or None if a slice is not uniform in 1/sigma (not a water field) or the library is missing."""
    pair = _conv_pair(which)
    if pair is None:
        return None
    xconv, yconv = pair
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    W, H, L = of.info["ray_dims"]
    S = int(beam.tracerSteps)
    idd = of.get("idd").reshape(L, S, H, W)
    rs = of.get("rsigma").reshape(L, S, H, W)
    tr = of.get("tile_radius").reshape(L, S, H // 8, W // 32)
    plan = of.get("layer_plan").reshape(L, 8)
    eff = of.get("eff_radius").reshape(L, -1)
    first = of.info["beam_first_inside"]
    bev = np.zeros((S, H + 64, W + 64), dtype=np.float32)
    seconds, n_slices, max_rad = 0.0, 0, 0
    for l in range(L):
        for k in range(first, int(plan[l, 6])):
            fin = np.isfinite(rs[l, k]) & (idd[l, k] != 0.0)
            if not fin.any():
                continue
            vals = rs[l, k][fin]
            if float(vals.max()) > float(vals.min()) * (1.0 + 1e-6):
                return None                                           # per-ray sigma: not separable
            cls = tr[l, k][tr[l, k] != 0xFF]
            rad = int(eff[l, int(cls.max())])
            src = np.ascontiguousarray(np.where(fin, idd[l, k], 0.0).astype(np.float32))
            ow, oh = W + 2 * rad, H + 2 * rad
            for _ in range(repeat):
                tmp = np.zeros((H, ow), dtype=np.float32)
                out = np.zeros((oh, ow), dtype=np.float32)
                t0 = time.perf_counter()
                xconv(P(src), P(tmp), C.c_float(float(vals.min())), rad, W, ow, H, rad)
                yconv(P(tmp), P(out), C.c_float(float(vals.min())), rad, H, ow, rad)
                seconds += time.perf_counter() - t0
            bev[k, 32 - rad:32 + H + rad, 32 - rad:32 + W + rad] += out
            n_slices += 1
            max_rad = max(max_rad, rad)
    return bev, seconds / repeat, n_slices, max_rad
