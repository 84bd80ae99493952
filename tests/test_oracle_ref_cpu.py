"""The oracle's superposition stage against the reference's OWN CPU convolution code on the reference's CPU-runnable configuration
(BASELINE.json configs[0]: water cube 128^3, one G000 field, one energy layer). In water a BEV slice has one sigma, so the
per-voxel-sigma superposition equals xConvCpu + yConvCpu of the slice (src/cpu_convolution_1d.cpp) — compiled from the reference's
sources into oracle/_ref/libref.so by oracle/Makefile. Skipped where that library is absent (no /root/reference at build time);
the oracle's restatement of the two routines (pinned bit for bit by golden vector G7) is always checked the same way."""
import numpy as np
import pytest

from raytracedicom_amd import scenarios


@pytest.mark.parametrize("which", ["port", "reference"])
def test_superposition_in_water_equals_the_separable_cpu_convolution(orc, synth, which):
    from oracle import ref_cpu_path
    if which == "reference" and ref_cpu_path.ref_lib() is None:
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    scn = scenarios.water_cube(synth, n=128, n_layers=1)
    dose = np.zeros_like(scn.ct)
    of = orc.run_field(scn, scn.beams[0], dose, keep_layers=True)
    assert of.status == 0
    r = ref_cpu_path.separable_bev(of, scn.beams[0], which=which)
    assert r is not None, "a water slice was not uniform in 1/sigma"
    sep, seconds, n_slices, max_rad = r
    W, H, L = of.info["ray_dims"]
    bev = of.get("bev").reshape(-1, H + 64, W + 64)
    assert n_slices > 50 and max_rad >= 2 and bev.max() > 0
    # same weights (erf differences), same products; the sums are formed in a different order (x then y against 2-D patches)
    err = np.abs(sep[:bev.shape[0]].astype(np.float64) - bev)
    assert err.max() <= 2e-6 * bev.max(), (err.max(), bev.max())
    big = bev > 1e-3 * bev.max()
    assert (err[big] / bev[big]).max() <= 1e-5
