"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN HOST CODE (oracle/_ref, built by
oracle/Makefile from /root/reference/src where it lies). Run in the build container only:

    make -C oracle && python oracle/make_golden.py

The fixtures are data (inputs + the reference's outputs); no reference source is stored.
G1  LUT text parser          energy_reader.cpp:12-101        -> golden_g1_lut_parse.npz
G2  search / interpolation   vector_find.h, vector_interpolate.h -> golden_g2_find_interp.npz
G7  erf-difference conv      cpu_convolution_1d.cpp:36-61,63-89,145-171 -> golden_g7_cpu_conv.npz
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF_LUTS = "/root/reference/LUTs/"
fp = C.POINTER(C.c_float)


def P(a):
    return a.ctypes.data_as(fp)


def load(name):
    L = C.CDLL(os.path.join(HERE, "_ref", name))
    L.ref_find_max.restype = C.c_float
    L.ref_find_decimal_ordered.restype = C.c_float
    L.ref_vector_interpolate.restype = C.c_float
    L.ref_energy_size.restype = C.c_long
    return L


def parse(L, directory):
    assert L.ref_energy_reader(directory.encode()) == 0, directory
    arrs = []
    for w in range(7):
        n = L.ref_energy_size(w)
        a = np.empty(n, dtype=np.float32)
        L.ref_energy_copy(w, P(a))
        arrs.append(a)
    ints = [C.c_int() for _ in range(4)]
    fl = [C.c_float() for _ in range(3)]
    L.ref_energy_scalars(C.byref(ints[0]), C.byref(ints[1]), C.byref(ints[2]), C.byref(fl[0]), C.byref(ints[3]),
                         C.byref(fl[1]), C.byref(ints[3]), C.byref(fl[2]))
    # the call above reuses ints[3] for nSp then nRRl; re-read cleanly
    a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    e = C.c_int()
    f0, f1, f2 = C.c_float(), C.c_float(), C.c_float()
    L.ref_energy_scalars(C.byref(a), C.byref(b), C.byref(c), C.byref(f0), C.byref(d), C.byref(f1), C.byref(e), C.byref(f2))
    scal = np.array([a.value, b.value, c.value, d.value, e.value], dtype=np.int64)
    scales = np.array([f0.value, f1.value, f2.value], dtype=np.float32)
    return arrs, scal, scales


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def g1():
    out = {}
    # (a) a SMALL synthetic LUT directory in the reference text layout, committed as fixture input
    from raytracedicom_amd import luts
    small_dir = os.path.join(GOLD, "lut_small")
    es = luts.synth_luts(n_energies=5, n_samples=16, n_hu=12)
    # perturb the water variant so the two radiation-length files differ like the reference's (index 2)
    luts.write_lut_dir(small_dir, es)
    rv = es.rRlVector.copy(); rv[2] = np.float32(rv[2] * 1.07)
    with open(os.path.join(small_dir, luts.FILES["rrl_water"]), "w") as fh:
        fh.write("%d %r\n\n" % (rv.size, float(np.float32(es.rRlScaleFact))))
        fh.write(" ".join(repr(float(x)) for x in rv) + "\n")
    for tag, libname in (("small", "libref.so"), ("small_water", "libref_water.so")):
        arrs, scal, scales = parse(load(libname), small_dir + "/")
        for i, a in enumerate(arrs):
            out["%s_arr%d" % (tag, i)] = a
        out[tag + "_scal"] = scal
        out[tag + "_scales"] = scales
    # (b) the reference's real tables: sizes, checksums and selected entries only (the data itself is not copied)
    for tag, libname in (("real", "libref.so"), ("real_water", "libref_water.so")):
        arrs, scal, scales = parse(load(libname), REF_LUTS)
        out[tag + "_scal"] = scal
        out[tag + "_scales"] = scales
        out[tag + "_sha256"] = np.array([sha(a) for a in arrs])
        out[tag + "_sum64"] = np.array([a.astype(np.float64).sum() for a in arrs])
        out[tag + "_head"] = np.stack([a[:4] for a in arrs])
        out[tag + "_tail"] = np.stack([a[-4:] for a in arrs])
        out[tag + "_rrl_1000_1004"] = arrs[6][1000:1004]
    np.savez(os.path.join(GOLD, "golden_g1_lut_parse.npz"), **out)
    return arrs  # real_water arrays


def g1n():
    """G1n: the NUCLEAR_CORR part of the LUT parser (energy_reader.cpp:103-162), one reference build per variant."""
    from raytracedicom_amd import luts
    out = {}
    small_dir = os.path.join(GOLD, "lut_small_nuc")
    es = luts.synth_luts(n_energies=5, n_samples=16, n_hu=12, nuclear=True)
    luts.write_lut_dir(small_dir, es)
    # make the three variants' files differ (the reference ships three different tables)
    for k, name in luts.NUC_FILES.items():
        es_k = luts.synth_luts(n_energies=5, n_samples=16, n_hu=12, nuclear=True)
        es_k.nucWeightMatrix *= np.float32(1.0 + 0.1 * k)
        es_k.nucSqSigmaMatrix += np.float32(k)
        with open(os.path.join(small_dir, name), "w") as fh:
            row = lambda a: " ".join(repr(float(x)) for x in np.asarray(a, dtype=np.float32))
            fh.write("%d %d\n\n" % (es.nEnergySamples, es.nEnergies))
            fh.write(row(es.energiesPerU) + "\n\n" + row(es.peakDepths) + "\n\n" + row(es.scaleFacts) + "\n\n")
            for r in es_k.nucWeightMatrix:
                fh.write(row(r) + "\n")
            fh.write("\n")
            for r in es_k.nucSqSigmaMatrix:
                fh.write(row(r) + "\n")
    for k, v in ((1, "SOUKUP"), (2, "FLUKA"), (3, "GAUSS_FIT")):
        L = load("libref_nuc_%s.so" % v)
        for tag, d in (("small", small_dir + "/"), ("real", REF_LUTS)):
            assert L.ref_energy_reader(d.encode()) == 0, d
            for w, nm in ((7, "weight"), (8, "sqsigma")):
                n = L.ref_energy_size(w)
                a = np.empty(n, dtype=np.float32)
                L.ref_energy_copy(w, P(a))
                if tag == "small":
                    out["small_%d_%s" % (k, nm)] = a
                else:
                    out["real_%d_%s_n" % (k, nm)] = np.int64(n)
                    out["real_%d_%s_sha256" % (k, nm)] = np.array(sha(a))
                    out["real_%d_%s_sum64" % (k, nm)] = a.astype(np.float64).sum()
                    out["real_%d_%s_head" % (k, nm)] = a[:4]
                    out["real_%d_%s_tail" % (k, nm)] = a[-4:]
    np.savez(os.path.join(GOLD, "golden_g1n_nuclear_lut.npz"), **out)


def g2(real):
    L = load("libref.so")
    energies, peaks, scales = real[0], real[1], real[2]
    out = {"energiesPerU": energies, "peakDepths": peaks, "scaleFacts": scales}
    # the 20 water-cube energies of main.cu:86-99, accumulated in float like the reference
    cur = np.float32(118.12); last = np.float32(172.51)
    step = np.float32((last - cur) / np.float32(19))
    q = []
    for _ in range(20):
        q.append(cur); cur = np.float32(cur + step)
    q = np.array(q, dtype=np.float32)
    extra = np.array([0.0, 62.3866, 62.38, 226.638, 300.0, energies[10], np.nextafter(energies[10], np.float32(0)),
                      np.nextafter(energies[10], np.float32(1e9)), 100.0, 150.5], dtype=np.float32)
    q = np.concatenate([q, extra])
    dec = np.array([L.ref_find_decimal_ordered(P(energies), energies.size, C.c_float(v)) for v in q], dtype=np.float32)
    out["query_energy"] = q
    out["decimal_idx"] = dec
    out["peak_interp"] = np.array([L.ref_vector_interpolate(P(peaks), peaks.size, C.c_float(v)) for v in dec], dtype=np.float32)
    out["scale_interp"] = np.array([L.ref_vector_interpolate(P(scales), scales.size, C.c_float(v)) for v in dec], dtype=np.float32)
    idxq = np.array([-1.0, 0.0, 0.25, 1.0, 35.999, 36.0, 145.5, 146.0, 146.5, 1000.0], dtype=np.float32)
    out["interp_query_idx"] = idxq
    out["interp_peaks"] = np.array([L.ref_vector_interpolate(P(peaks), peaks.size, C.c_float(v)) for v in idxq], dtype=np.float32)
    # ordered lists with duplicates and plateaus (like a min-WEPL curve) for the first-larger / last-smaller searches
    rng = np.random.default_rng(5)
    lists, vals, fl, ls, mx = [], [], [], [], []
    for n in (1, 2, 3, 7, 64, 512):
        a = np.sort(rng.random(n).astype(np.float32) * 300).astype(np.float32)
        if n >= 7:
            a[n // 2] = a[n // 2 - 1]            # duplicate
            a[-2] = a[-1]
        for v in list(a[:: max(1, n // 5)]) + [np.float32(-1.0), np.float32(1e6), np.float32(a[0] + 0.001), a[-1]]:
            v = np.float32(v)
            lists.append(a); vals.append(v)
            fl.append(L.ref_find_first_larger_ordered(P(a), n, C.c_float(v)))
            ls.append(L.ref_find_last_smaller_or_eq_ordered(P(a), n, C.c_float(v)))
            mx.append(L.ref_find_max(P(a), n))
    out["search_lists"] = np.array([np.pad(a, (0, 512 - a.size), constant_values=np.nan) for a in lists], dtype=np.float32)
    out["search_n"] = np.array([a.size for a in lists], dtype=np.int32)
    out["search_val"] = np.array(vals, dtype=np.float32)
    out["first_larger"] = np.array(fl, dtype=np.int32)
    out["last_smaller_eq"] = np.array(ls, dtype=np.int32)
    out["find_max"] = np.array(mx, dtype=np.float32)
    np.savez(os.path.join(GOLD, "golden_g2_find_interp.npz"), **out)


def g7():
    L = load("libref.so")
    rng = np.random.default_rng(11)
    out = {}
    cases = [(0.35, 6), (0.12, 16), (1.7, 1), (0.7071, 3), (0.05, 32), (3.0, 0)]
    out["cases"] = np.array(cases, dtype=np.float32)
    for ci, (rs, rad) in enumerate(cases):
        rad = int(rad)
        inW, H = 23, 9
        a = rng.random((H, inW), dtype=np.float32) * 100
        # xConvCpu gather with output wider than input
        outW = inW + 2 * rad
        xo = np.zeros((H, outW), dtype=np.float32)
        L.ref_x_conv_cpu(P(a), P(xo), C.c_float(rs), rad, inW, outW, H, rad)
        # xConvCpuScat scatter
        xs = np.zeros((H, outW), dtype=np.float32)
        L.ref_x_conv_cpu_scat(P(a), P(xs), C.c_float(rs), rad, inW, outW, H, rad)
        # yConvCpu scatter
        yo = np.zeros((H + 2 * rad, inW), dtype=np.float32)
        L.ref_y_conv_cpu(P(a), P(yo), C.c_float(rs), rad, H, inW, rad)
        out["in%d" % ci] = a
        out["xgather%d" % ci] = xo
        out["xscatter%d" % ci] = xs
        out["yscatter%d" % ci] = yo
    np.savez(os.path.join(GOLD, "golden_g7_cpu_conv.npz"), **out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "g1n":
    g1n()
    print("wrote golden_g1n_nuclear_lut.npz")
    sys.exit(0)

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    real = g1()
    g2(real)
    g7()
    print("golden fixtures written to", GOLD)
