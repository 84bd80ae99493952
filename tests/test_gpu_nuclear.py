"""NUCLEAR_CORR (default off in the reference, CMakeLists.txt:61-69), restated as the reference's code
behaves — including its nuclear memory step of 0 (kernel_wrapper.cu:925), which makes the halo reach the dose through BEV slice
0 only, i.e. only when the beam starts inside the patient. The HIP engine against the oracle for the three variants."""
import numpy as np
import pytest

from raytracedicom_amd import abi, luts, scenarios

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nuc_luts():
    return luts.synth_luts(nuclear=True)


def _run(orc, engine, scn, opt):
    ref = np.zeros_like(scn.ct)
    of = orc.run_field(scn, scn.beams[0], ref, options=opt, keep_layers=True)
    assert of.status == 0, of.error
    dose = np.zeros_like(scn.ct)
    with engine.Engine(0) as eng:
        eng.set_options(opt)
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        n = scn.n_voxels
        d = eng.device_alloc(4 * n)
        eng.device_zero(d, 4 * n)
        f = eng.create_field(scn.beams[0], scn.dims)
        f.compute(d)
        _, info = f.finish()
        eng.to_host(dose, d)
        W, H, L = of.info["ray_dims"]
        S = scn.beams[0].tracerSteps
        # the sigma chain with the variant's constants: radius classes stay bit-exact (over the steps the reference classifies)
        plan = of.get("layer_plan").reshape(L, 8)
        tr_g = f.fetch("tile_radius").reshape(L, S, H // 8, W // 32)
        tr_o = of.get("tile_radius").reshape(L, S, H // 8, W // 32)
        for l in range(L):
            a0, lfp = of.info["beam_first_inside"], int(plan[l, 6])
            np.testing.assert_array_equal(tr_g[l, a0:lfp], tr_o[l, a0:lfp])
        np.testing.assert_array_equal(f.fetch("eff_radius"), of.get("eff_radius"))
        idd_g, idd_o = f.fetch("idd").reshape(L, S, H, W), of.get("idd").reshape(L, S, H, W)
        for l in range(L):
            a0, a1 = of.info["beam_first_inside"], int(plan[l, 5])
            np.testing.assert_allclose(idd_g[l, a0:a1], idd_o[l, a0:a1], rtol=2e-5, atol=1e-12)
        f.destroy()
        eng.device_free(d)
    mx = float(ref.max())
    assert mx > 0
    thr = ref > 1e-3 * mx
    assert (np.abs(dose - ref)[thr] <= 1e-4 * ref[thr] + 1e-6 * mx).all(), float((np.abs(dose - ref)[thr] / ref[thr]).max())
    assert np.abs(dose - ref).max() <= 2e-5 * mx
    rate, n_eval, _ = orc.gamma_pass_rate(ref, dose, scn.spacing)
    assert rate == 1.0 and n_eval > 0
    return dose, ref, of.info


@pytest.mark.parametrize("variant", [abi.RTD_NUC_SOUKUP, abi.RTD_NUC_FLUKA, abi.RTD_NUC_GAUSS_FIT])
def test_nuclear_variants_beam_starting_inside_the_patient(orc, engine, nuc_luts, variant):
    """The reference's water cube (tracer starts inside the cube: entry step 0): the primary dose loses the nuclear fraction and
    slice 0 of the halo cube is deposited."""
    scn = scenarios.water_cube(nuc_luts, n=64, n_layers=3, spots=7, pitch=6.0)
    opt = abi.default_options()
    opt.nuclear_corr = variant
    dose, ref, info = _run(orc, engine, scn, opt)
    assert info["beam_first_inside"] == 0
    base = np.zeros_like(scn.ct)
    orc.run_field(scn, scn.beams[0], base, keep_layers=False).close()
    assert ref.sum() < base.sum()                                     # the nuclear fraction left the primary dose
    assert ref.sum() > 0.9 * base.sum()


@pytest.mark.parametrize("variant", [abi.RTD_NUC_FLUKA])
def test_nuclear_beam_entering_from_outside(orc, engine, nuc_luts, variant):
    """Heterogeneous phantom, the beam enters from air (entry step > 0): the halo cube stays empty (the reference's nuclear arrays
    are only ever written in plane 0), the primary is scaled."""
    ct, _ = scenarios.hetero_phantom(96)
    beam = scenarios.make_field(nuc_luts, 96, 256.0 / 96, (-128.0, -128.0, -106.0), 0.0, 5, 7.0, 3, 11, start_z=150.0)   # 23 mm of air first
    scn = scenarios.Scenario("hetero96_air_gap", nuc_luts, ct, (256.0 / 96,) * 3, [beam])
    opt = abi.default_options()
    opt.nuclear_corr = variant
    dose, ref, info = _run(orc, engine, scn, opt)
    assert info["beam_first_inside"] > 0


def test_nuclear_needs_its_tables(engine, synth):
    scn = scenarios.water_cube(synth, n=32, n_layers=1, spots=3)
    opt = abi.default_options()
    opt.nuclear_corr = abi.RTD_NUC_SOUKUP
    with engine.Engine(0) as eng:
        eng.set_options(opt)
        eng.set_luts(synth)                                           # no nuclear tables
        eng.set_ct(scn.ct)
        with pytest.raises(engine.RtdError) as e:
            eng.create_field(scn.beams[0], scn.dims)
        assert e.value.status == abi.RTD_ERR_INVALID_ARG and "nuclear" in str(e.value)


def test_nuclear_lut_directory_loader(orc, engine, nuc_luts, tmp_path):
    """rtd_load_luts_dir reads the variant's nuclear file when options.nuclear_corr is set (energy_reader.cpp:103-162)."""
    d = str(tmp_path / "luts")
    luts.write_lut_dir(d, nuc_luts)
    scn = scenarios.water_cube(nuc_luts, n=48, n_layers=2, spots=5, pitch=6.0)
    opt = abi.default_options()
    opt.nuclear_corr = abi.RTD_NUC_GAUSS_FIT
    doses = []
    for mode in ("dir", "arrays"):
        with engine.Engine(0) as eng:
            eng.set_options(opt)
            if mode == "dir":
                eng.load_luts_dir(d, True)
            else:
                eng.set_luts(luts.read_lut_dir(d, True, nuclear_corr=abi.RTD_NUC_GAUSS_FIT))
            eng.set_ct(scn.ct)
            dose = np.zeros_like(scn.ct)
            eng.compute(scn.beams, dose)
            doses.append(dose)
    assert doses[0].max() > 0
    np.testing.assert_array_equal(doses[0], doses[1])
