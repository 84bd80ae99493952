"""The oracle against the golden vectors produced by the REFERENCE's own host code (oracle/make_golden.py):
G1 LUT parser (energy_reader.cpp), G2 search/interpolation (vector_find.h, vector_interpolate.h),
G7 erf-difference convolution weights (cpu_convolution_1d.cpp). Bit-exact comparisons."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, REF_LUTS
from raytracedicom_amd import abi, luts

fp = abi.c_float_p


def P(a):
    return a.ctypes.data_as(fp)


def _oracle_parse(orc, directory, water):
    L = orc.lib()
    s = abi.RtdLuts()
    assert L.orc_read_luts(directory.encode(), int(water), C.byref(s)) == 0
    nE, nS = s.n_energies, s.n_energy_samples
    arrs = [np.ctypeslib.as_array(s.energies_per_u, (nE,)).copy(), np.ctypeslib.as_array(s.peak_depths, (nE,)).copy(),
            np.ctypeslib.as_array(s.scale_facts, (nE,)).copy(), np.ctypeslib.as_array(s.cidd_matrix, (nE * nS,)).copy(),
            np.ctypeslib.as_array(s.density_vector, (s.n_density_samples,)).copy(),
            np.ctypeslib.as_array(s.sp_vector, (s.n_sp_samples,)).copy(),
            np.ctypeslib.as_array(s.rrl_vector, (s.n_rrl_samples,)).copy()]
    scal = [nS, nE, s.n_density_samples, s.n_sp_samples, s.n_rrl_samples]
    scales = [s.density_scale_fact, s.sp_scale_fact, s.rrl_scale_fact]
    L.orc_luts_free(C.byref(s))
    return arrs, scal, scales


def _python_parse(directory, water):
    es = luts.read_lut_dir(directory, water)
    arrs = [es.energiesPerU, es.peakDepths, es.scaleFacts, es.ciddMatrix.reshape(-1), es.densityVector, es.spVector,
            es.rRlVector]
    scal = [es.nEnergySamples, es.nEnergies, es.nDensitySamples, es.nSpSamples, es.nRRlSamples]
    scales = [es.densityScaleFact, es.spScaleFact, es.rRlScaleFact]
    return arrs, scal, scales


@pytest.mark.parametrize("water", [False, True])
@pytest.mark.parametrize("parser", ["oracle", "python"])
def test_g1_small_lut_dir(orc, water, parser):
    g = np.load(os.path.join(GOLDEN, "golden_g1_lut_parse.npz"))
    tag = "small_water" if water else "small"
    d = os.path.join(GOLDEN, "lut_small") + "/"
    arrs, scal, scales = _oracle_parse(orc, d, water) if parser == "oracle" else _python_parse(d, water)
    assert list(scal) == list(g[tag + "_scal"])
    np.testing.assert_array_equal(np.array(scales, dtype=np.float32), g[tag + "_scales"])
    for i, a in enumerate(arrs):
        np.testing.assert_array_equal(a, g["%s_arr%d" % (tag, i)])
    if water:  # the two radiation-length files differ in exactly one entry, like the reference's
        assert (g["small_arr6"] != g["small_water_arr6"]).sum() == 1


@pytest.mark.skipif(not os.path.isdir(REF_LUTS), reason="reference LUT files only exist in the build container")
@pytest.mark.parametrize("water", [False, True])
@pytest.mark.parametrize("parser", ["oracle", "python"])
def test_g1_real_lut_dir(orc, water, parser):
    g = np.load(os.path.join(GOLDEN, "golden_g1_lut_parse.npz"))
    tag = "real_water" if water else "real"
    arrs, scal, scales = _oracle_parse(orc, REF_LUTS, water) if parser == "oracle" else _python_parse(REF_LUTS, water)
    assert list(scal) == list(g[tag + "_scal"]) == [1024, 147, 3072, 3072, 3072]
    np.testing.assert_array_equal(np.array(scales, dtype=np.float32), g[tag + "_scales"])
    for i, a in enumerate(arrs):
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(g[tag + "_sha256"][i])
        np.testing.assert_array_equal(a[:4], g[tag + "_head"][i])
        np.testing.assert_array_equal(a[-4:], g[tag + "_tail"][i])


def test_g2_decimal_and_interpolate(orc):
    g = np.load(os.path.join(GOLDEN, "golden_g2_find_interp.npz"))
    L = orc.lib()
    e, p, s = (abi.f32(g[k]) for k in ("energiesPerU", "peakDepths", "scaleFacts"))
    dec = np.array([L.orc_find_decimal_ordered(P(e), e.size, C.c_float(v)) for v in g["query_energy"]], dtype=np.float32)
    np.testing.assert_array_equal(dec, g["decimal_idx"])
    pk = np.array([L.orc_vector_interpolate(P(p), p.size, C.c_float(v)) for v in dec], dtype=np.float32)
    sc = np.array([L.orc_vector_interpolate(P(s), s.size, C.c_float(v)) for v in dec], dtype=np.float32)
    np.testing.assert_array_equal(pk, g["peak_interp"])
    np.testing.assert_array_equal(sc, g["scale_interp"])
    ip = np.array([L.orc_vector_interpolate(P(p), p.size, C.c_float(v)) for v in g["interp_query_idx"]], dtype=np.float32)
    np.testing.assert_array_equal(ip, g["interp_peaks"])
    # 118.12 MeV/u -> idx 36.0, peak 100.799 mm, scale 8.12509
    assert dec[0] == 36.0 and abs(pk[0] - 100.799) < 1e-4 and abs(sc[0] - 8.12509) < 1e-5


def test_g2_ordered_searches(orc):
    g = np.load(os.path.join(GOLDEN, "golden_g2_find_interp.npz"))
    L = orc.lib()
    for lst, n, v, fl, ls, mx in zip(g["search_lists"], g["search_n"], g["search_val"], g["first_larger"],
                                     g["last_smaller_eq"], g["find_max"]):
        a = abi.f32(lst[:n])
        assert L.orc_find_first_larger_ordered(P(a), int(n), C.c_float(v)) == fl
        assert L.orc_find_last_smaller_or_eq_ordered(P(a), int(n), C.c_float(v)) == ls
        assert np.float32(L.orc_find_max(P(a), int(n))) == mx


def test_g7_cpu_convolution_weights(orc):
    """Oracle's separable convolution == reference's xConvCpu / xConvCpuScat / yConvCpu, bit for bit; and the
    kernel-superposition weights (kernel_wrapper.cuh:459-467) equal those weights except entry 0's formula."""
    g = np.load(os.path.join(GOLDEN, "golden_g7_cpu_conv.npz"))
    L = orc.lib()
    for ci, (rs, rad) in enumerate(g["cases"]):
        rad = int(rad)
        a = abi.f32(g["in%d" % ci])
        H, inW = a.shape
        outW = inW + 2 * rad
        xo = np.zeros((H, outW), dtype=np.float32)
        L.orc_x_conv_cpu(P(a), P(xo), C.c_float(rs), rad, inW, outW, H, rad)
        np.testing.assert_array_equal(xo, g["xgather%d" % ci])
        xs = np.zeros((H, outW), dtype=np.float32)
        L.orc_x_conv_cpu_scat(P(a), P(xs), C.c_float(rs), rad, inW, outW, H, rad)
        np.testing.assert_array_equal(xs, g["xscatter%d" % ci])
        yo = np.zeros((H + 2 * rad, inW), dtype=np.float32)
        L.orc_y_conv_cpu(P(a), P(yo), C.c_float(rs), rad, H, inW, rad)
        np.testing.assert_array_equal(yo, g["yscatter%d" % ci])
        # scatter and gather of the reference agree to rounding (different summation order) right of the columns
        # that the gather's unsigned index arithmetic leaves at zero (cpu_convolution_1d.cpp:53)
        np.testing.assert_allclose(g["xscatter%d" % ci][:, rad:], g["xgather%d" % ci][:, rad:], rtol=2e-6, atol=1e-6)
        # KS weights: impulse response of the reference convolution == orc_erf_diffs
        e = np.zeros(rad + 1, dtype=np.float32)
        L.orc_erf_diffs(C.c_float(rs), rad, P(e))
        imp = np.zeros((1, 2 * rad + 1), dtype=np.float32)
        imp[0, rad] = 1.0
        resp = np.zeros((1, 2 * rad + 1), dtype=np.float32)
        L.orc_x_conv_cpu(P(imp), P(resp), C.c_float(rs), rad, 2 * rad + 1, 2 * rad + 1, 1, 0)
        np.testing.assert_array_equal(resp[0, rad + 1:], e[1:])          # same expression for i >= 1
        np.testing.assert_allclose(resp[0, rad], e[0], rtol=1.2e-7)      # erf(r/2) vs 0.5*(erf(r/2)+erf(r/2))


# ---- G1n: NUCLEAR_CORR tables (energy_reader.cpp:103-162), pinned by one reference build per variant -------------------
@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("parser", ["oracle", "python"])
def test_g1n_nuclear_tables_small_dir(orc, variant, parser):
    g = np.load(os.path.join(GOLDEN, "golden_g1n_nuclear_lut.npz"))
    d = os.path.join(GOLDEN, "lut_small_nuc") + "/"
    if parser == "oracle":
        L = orc.lib()
        s = abi.RtdLuts()
        assert L.orc_read_luts_nuc(d.encode(), 0, variant, C.byref(s)) == 0
        n = s.n_energies * s.n_energy_samples
        w, q = np.ctypeslib.as_array(s.nuc_weight_matrix, (n,)).copy(), np.ctypeslib.as_array(s.nuc_sq_sigma_matrix, (n,)).copy()
        L.orc_luts_free(C.byref(s))
    else:
        es = luts.read_lut_dir(d, False, nuclear_corr=variant)
        w, q = es.nucWeightMatrix.reshape(-1), es.nucSqSigmaMatrix.reshape(-1)
    np.testing.assert_array_equal(w, g["small_%d_weight" % variant])
    np.testing.assert_array_equal(q, g["small_%d_sqsigma" % variant])


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_g1n_nuclear_tables_of_the_reference(orc, variant):
    """The reference's real nuclear tables (read where they lie; skipped where /root/reference is absent): sizes, checksums, ends."""
    if not os.path.isdir(REF_LUTS):
        pytest.skip("reference LUTs not present")
    g = np.load(os.path.join(GOLDEN, "golden_g1n_nuclear_lut.npz"))
    es = luts.read_lut_dir(REF_LUTS, False, nuclear_corr=variant)
    L = orc.lib()
    s = abi.RtdLuts()
    assert L.orc_read_luts_nuc(REF_LUTS.encode(), 0, variant, C.byref(s)) == 0
    n = s.n_energies * s.n_energy_samples
    for nm, py, oc in (("weight", es.nucWeightMatrix.reshape(-1), np.ctypeslib.as_array(s.nuc_weight_matrix, (n,)).copy()),
                       ("sqsigma", es.nucSqSigmaMatrix.reshape(-1), np.ctypeslib.as_array(s.nuc_sq_sigma_matrix, (n,)).copy())):
        for a in (py, oc):
            assert a.size == int(g["real_%d_%s_n" % (variant, nm)])
            assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(g["real_%d_%s_sha256" % (variant, nm)])
            np.testing.assert_array_equal(a[:4], g["real_%d_%s_head" % (variant, nm)])
            np.testing.assert_array_equal(a[-4:], g["real_%d_%s_tail" % (variant, nm)])
    L.orc_luts_free(C.byref(s))
