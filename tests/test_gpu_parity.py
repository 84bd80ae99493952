"""GPU parity tests: the HIP engine through its C ABI against the CPU oracle on the same seeded inputs.

Tolerances (stated per stage):
  * tracer outputs (density, WEPL, first inside/outside, min WEPL): bit-exact — same IEEE operations, no
    transcendentals;
  * spot->ray weights: bit-exact — the convolution is IEEE operations in the reference's order plus rtd_erf_det
    (include/rtd_detmath.h), the error function both sides evaluate; hence the RAY_WEIGHT_CUTOFF liveness of every ray is too;
  * IDD, 1/sigma: rtol 2e-5 — the per-ray values use the hardware reciprocal / sqrt (<= 2 ulp);
  * tile radius classes, batch radii (index work): bit-exact — every operation the class depends on is IEEE in the kernel,
    the reference's __powf being rtd_pow_det on both sides; the failure message prints the offending bits;
  * BEV dose, final dose: rtol 1e-4 on voxels above 1e-3 of the maximum (+ atol 1e-6*max) — the superposition sums the
    same terms in a different (fixed) order than the oracle and uses a Gaussian series for its weight tables;
    gamma(1 %/1 mm) >= 99 % is the north-star bar and is asserted at 100 %.
"""
import math

import numpy as np
import pytest

from raytracedicom_amd import abi, scenarios

pytestmark = pytest.mark.gpu


def _run_engine(engine, scn, beam, options=None):
    eng = engine.Engine(0)
    if options is not None:
        eng.set_options(options)
    eng.set_luts(scn.luts)
    eng.set_ct(scn.ct)
    n = scn.n_voxels
    d_dose = eng.device_alloc(4 * n)
    eng.device_zero(d_dose, 4 * n)
    fld = eng.create_field(beam, scn.dims)
    fld.compute(d_dose)
    timing, info = fld.finish()
    dose = np.empty_like(scn.ct)
    eng.to_host(dose, d_dose)
    return eng, fld, dose, timing, info, d_dose


def _rel_close(a, b, rtol, floor_frac=1e-3, atol_frac=1e-6):
    a = a.astype(np.float64); b = b.astype(np.float64)
    mx = np.abs(b).max()
    mask = np.abs(b) > floor_frac * mx
    err = np.abs(a - b)
    assert (err[mask] <= rtol * np.abs(b[mask]) + atol_frac * mx).all(), \
        "max rel err %g" % (err[mask] / np.abs(b[mask])).max()
    assert (err[~mask] <= 2 * rtol * floor_frac * mx + atol_frac * mx).all()


def _compare_field(orc, engine, scn, beam, options=None):
    dose_ref = np.zeros_like(scn.ct)
    of = orc.run_field(scn, beam, dose_ref, options=options, keep_layers=True)
    assert of.status == 0
    eng, fld, dose, timing, info, d_dose = _run_engine(engine, scn, beam, options)
    try:
        oi = of.info
        for k in ("ray_dims", "beam_first_inside", "beam_first_outside", "beam_first_guaranteed_passive",
                  "beam_first_calculated_passive", "bbox_min", "bbox_max", "live_steps", "max_radius"):
            assert info[k] == oi[k], (k, info[k], oi[k])
        np.testing.assert_array_equal(np.array(info["ray_offset"], np.float32), np.array(oi["ray_offset"], np.float32))
        W, H, L = oi["ray_dims"]
        S = beam.tracerSteps
        # stage 1: tracer — bit-exact
        for name in ("density", "wepl", "first_inside", "first_outside", "wepl_min"):
            np.testing.assert_array_equal(fld.fetch(name), of.get(name), err_msg=name)
        # stage 2: plan + spot->ray weights
        np.testing.assert_allclose(fld.fetch("layer_plan").reshape(L, 8)[:, :6], of.get("layer_plan").reshape(L, 8)[:, :6], rtol=1e-6)
        np.testing.assert_array_equal(fld.fetch("ray_weights"), of.get("ray_weights"))
        # stage 3: fill
        first, calc = oi["beam_first_inside"], oi["beam_first_calculated_passive"]
        np.testing.assert_array_equal(fld.fetch("first_passive"), of.get("first_passive"))
        plan = of.get("layer_plan").reshape(L, 8)
        idd_g, idd_o = fld.fetch("idd").reshape(L, S, H, W), of.get("idd").reshape(L, S, H, W)
        rs_g, rs_o = fld.fetch("rsigma").reshape(L, S, H, W), of.get("rsigma").reshape(L, S, H, W)
        tr_g = fld.fetch("tile_radius").reshape(L, S, H // 8, W // 32)
        tr_o = of.get("tile_radius").reshape(L, S, H // 8, W // 32)
        for l in range(L):
            a0, a1 = first, int(plan[l, 5])
            np.testing.assert_allclose(idd_g[l, a0:a1], idd_o[l, a0:a1], rtol=2e-5, atol=1e-12, err_msg="idd layer %d" % l)
            fin = np.isfinite(rs_o[l, a0:a1])
            np.testing.assert_array_equal(np.isfinite(rs_g[l, a0:a1]), fin)
            np.testing.assert_allclose(rs_g[l, a0:a1][fin], rs_o[l, a0:a1][fin], rtol=2e-5)
            lfp = int(plan[l, 6])
            # index work: radius class of every (step, tile) bit-exact (tileRadCalc, kernel_wrapper.cuh:256-313)
            if not np.array_equal(tr_g[l, a0:lfp], tr_o[l, a0:lfp]):
                ks, tys, txs = np.nonzero(tr_g[l, a0:lfp] != tr_o[l, a0:lfp])
                k, ty, tx = int(ks[0]) + a0, int(tys[0]), int(txs[0])
                mg = rs_g[l, k, 8 * ty:8 * ty + 8, 32 * tx:32 * tx + 32].min()
                mo = rs_o[l, k, 8 * ty:8 * ty + 8, 32 * tx:32 * tx + 32].min()
                raise AssertionError("tile_radius differs on %d (step, tile) of layer %d; first at step %d tile (%d, %d): engine %d, oracle %d; "
                                     "tile minimum of 1/sigma: engine %r (0x%08x), oracle %r (0x%08x)"
                                     % (ks.size, l, k, tx, ty, tr_g[l, k, ty, tx], tr_o[l, k, ty, tx], float(mg), np.float32(mg).view(np.uint32),
                                        float(mo), np.float32(mo).view(np.uint32)))
        # batch radius per radius class (host batching rule, kernel_wrapper.cu:966-976)
        np.testing.assert_array_equal(fld.fetch("eff_radius").reshape(L, -1), of.get("eff_radius").reshape(L, -1))
        # stage 4/5: BEV and final dose
        bev_g, bev_o = fld.fetch("bev"), of.get("bev")
        _rel_close(bev_g, bev_o, rtol=1e-4)
        _rel_close(dose, dose_ref, rtol=1e-4)
        rate, n_eval, gmax = orc.gamma_pass_rate(dose_ref, dose, scn.spacing)
        assert n_eval > 0 and rate == 1.0, (rate, n_eval, gmax)
        return dose, dose_ref, timing, info
    finally:
        fld.destroy()
        eng.device_free(d_dose)
        eng.close()


@pytest.fixture(scope="module")
def ct512():
    return scenarios.hetero_phantom(512)[0]


@pytest.fixture(scope="module")
def ct768():
    return scenarios.hetero_phantom(768)[0]


def test_c3_hetero_512_bench_workload(orc, engine, synth, ct512):
    """BASELINE.json configs[2] = the bench.py workload: 512^3 heterogeneous CT, one field, 10x10 spots x 20 layers. Every
    intermediate, the BEV dose, the dose and gamma against the oracle."""
    scn = scenarios.hetero_ct(synth, n=512, angles=[0.0], ct=ct512)
    dose, ref, timing, info = _compare_field(orc, engine, scn, scn.beams[0])
    assert info["ray_dims"] == [96, 88, 20] and info["live_steps"] > 3000
    assert info["uniform_sigma"] == 0                                # heterogeneous CT: per-voxel-sigma superposition


@pytest.mark.parametrize("deg", [90.0, 180.0, 270.0])
def test_c4_fields_of_the_four_angle_plan(orc, engine, synth, ct512, deg):
    """BASELINE.json configs[3], field by field at full size: the along-beam tracer (k_trace_sample_t) and the transposed
    transfer (k_transfer_t) run at 90 / 270 degrees."""
    scn = scenarios.hetero_ct(synth, n=512, angles=[0.0, 90.0, 180.0, 270.0], ct=ct512)
    _compare_field(orc, engine, scn, scn.beams[int(deg // 90)])


def test_c4_four_field_plan_sum(orc, engine, synth, ct512):
    """BASELINE.json configs[3] as one reference-shaped call: the four fields accumulated into one volume."""
    scn = scenarios.hetero_ct(synth, n=512, angles=[0.0, 90.0, 180.0, 270.0], ct=ct512)
    ref = orc.compute(scn)
    dose = np.zeros_like(scn.ct)
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        eng.compute(scn.beams, dose)
    _rel_close(dose, ref, rtol=1e-4)
    rate, n_eval, gmax = orc.gamma_pass_rate(ref, dose, scn.spacing)
    assert rate == 1.0 and n_eval > 1000000, (rate, n_eval, gmax)


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4, 5, 6, 7])
def test_c5_hetero_768_fields(orc, engine, synth, ct768, idx):
    """BASELINE.json configs[4]: 768^3 CT (voxel 1/3 mm), every field of the eight-angle plan (0, 45, ..., 315 degrees)."""
    scn = scenarios.hetero_ct(synth, n=768, n_fields=8, ct=ct768)
    _compare_field(orc, engine, scn, scn.beams[idx])


def test_c1_water_cube_128_single_layer(orc, engine, synth):
    """BASELINE.json configs[0]: water cube 128^3, single G000 field, one energy layer."""
    scn = scenarios.water_cube(synth, n=128, n_layers=1)
    dose, ref, timing, info = _compare_field(orc, engine, scn, scn.beams[0])
    assert dose.max() > 0 and timing["total_ms"] > 0


def test_water_cube_multi_layer_fine_timing(orc, engine, synth):
    """Several layers + FINE_GRAINED_TIMING buckets."""
    scn = scenarios.water_cube(synth, n=128, n_layers=4, spots=17, pitch=4.0)
    opt = abi.default_options()
    opt.fine_grained_timing = 1
    dose, ref, timing, info = _compare_field(orc, engine, scn, scn.beams[0], options=opt)
    parts = sum(timing[k] for k in ("raytracing_ms", "prepare_energy_loop_ms", "fill_idd_sigma_ms", "prepare_superp_ms",
                                    "superp_ms", "transforming_ms"))
    assert parts == pytest.approx(timing["total_ms"], rel=0.05)


@pytest.mark.parametrize("deg,dist", [(0.0, (math.inf, math.inf)), (90.0, (math.inf, math.inf)), (37.0, (2000.0, 2500.0)),
                                      (180.0, (1800.0, 1800.0))])
def test_heterogeneous_rotated_divergent(orc, engine, synth, deg, dist):
    """Heterogeneous phantom (air gap, lung, bone, cavity), rotated gantry, finite source distance."""
    ct, _ = scenarios.hetero_phantom(128)
    scn = scenarios.hetero_ct(synth, n=128, spots=6, pitch=7.0, n_layers=3, angles=[deg], source_dist=dist, ct=ct)
    _compare_field(orc, engine, scn, scn.beams[0])


@pytest.mark.parametrize("rot,steps", [(((0, 0, 1), (1, 0, 0), (0, 1, 0)), 300), (((0, 0, -1), (0.8, -0.6, 0), (-0.6, -0.8, 0)), 300),
                                       (((0.8, 0, 0.6), (0, 1, 0), (-0.6, 0, 0.8)), 300),
                                       (((0, 0, 1), (1, 0, 0), (0, 1, 0)), 600)])      # > 512 steps: two passes of the along-beam tracer
def test_beam_axes_other_than_rotation_about_y(orc, engine, synth, rot, steps):
    """Gantry frames whose BEV x axis runs along dose y (first two: the beam runs along the CT x axis, so the tracer walks one ray
    per wave and the transfer lays its lanes along y) and an oblique one: the orientation-specific kernels against the oracle."""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=5, pitch=7.0, n_layers=3, angles=[0.0], source_dist=(1500.0, 2100.0), steps=steps, ct=ct,
                              gantry_rot=rot)
    _compare_field(orc, engine, scn, scn.beams[0])


@pytest.mark.parametrize("dist", [(math.inf, math.inf), (1400.0, 1900.0)])
def test_beam_along_dose_y_fine_voxels(orc, engine, synth, dist):
    """The BEV depth axis maps to dose y (gantry z = world +y: the beam travels towards -y) on voxels finer than the step
    (384^3 on the 256 mm cube: 2/3 mm voxels, 1 mm steps). The reference's transfer launch rounds its grid up to whole 32 x 8 blocks
    and masks by the dose dimensions only (kernel_wrapper.cu:1209-1210, guard :80), so y, like x, runs past maxIdx.y up to the block
    edge; here that edge lies in front of the entry slice, where the BEV value one voxel beyond maxIdx.y still interpolates against
    the first slice. Engine and oracle must both deposit there (the oracle stopped at maxIdx.y until round 3)."""
    n = 384
    ct, voxel = scenarios.hetero_phantom(n)
    rot = ((1, 0, 0), (0, 0, 1), (0, -1, 0))                          # gantry x -> world x, y -> -z, z -> +y
    # the beam starts INSIDE the volume (world y = 60.633 mm = voxel 282.95), so that rows past maxIdx.y = 283 exist, and voxel
    # 284 lies 0.7 steps in front of the entry slice: the launch's over-run rows receive ~12 % of the maximum dose
    beam = scenarios.make_field(synth, n, voxel, (-128.0, -128.0, -106.0), 0.0, 5, 7.0, 3, 11, dist, 300, start_z=60.633, gantry_rot=rot)
    scn = scenarios.Scenario("beam along -y", synth, ct, (voxel,) * 3, [beam])
    dose, ref, timing, info = _compare_field(orc, engine, scn, scn.beams[0])
    ymax = info["bbox_max"][1]
    cov = min(info["bbox_min"][1] + ((ymax - info["bbox_min"][1] + 1 + 7) // 8) * 8 - 1, scn.dims[1] - 1)
    assert cov > ymax                                                 # the launch does run past maxIdx.y in this geometry
    over = ref[:, ymax + 1:cov + 1, :]                                # [z][y][x]
    assert over.max() > 0.05 * ref.max(), "the rows past maxIdx.y carry no dose: the case is not exercised"
    np.testing.assert_array_equal(dose[:, ymax + 1:cov + 1, :] > 0, over > 0)


def test_spot_map_taller_than_the_lds_tile(orc, engine, synth):
    """A spot map of 400 rows (> kConvMaxRows = 384): the spot -> ray convolution runs as the two launches k_conv_x / k_conv_y through
    the intermediate buffer instead of the fused k_conv (gpu_convolution_2d.cu:16-59 is two launches as well); ray weights bit-exact."""
    ct, _ = scenarios.hetero_phantom(64)
    scn = scenarios.hetero_ct(synth, n=64, spots=(3, 400), pitch=0.6, n_layers=2, angles=[0.0], steps=120, ct=ct)
    _, _, _, info = _compare_field(orc, engine, scn, scn.beams[0])
    assert info["ray_dims"][1] >= 240


def test_c3_through_the_output_stationary_kernel(orc, engine, synth, ct512, monkeypatch):
    """The bench field with the row sweep switched off (RTD_NO_SWEEP, read at field creation): k_superpose_mfma, round 2's kernel — kept
    as a second implementation of the superposition that the sweep is compared with — keeps its full-size parity evidence."""
    monkeypatch.setenv("RTD_NO_SWEEP", "1")
    scn = scenarios.hetero_ct(synth, n=512, angles=[0.0], ct=ct512)
    _compare_field(orc, engine, scn, scn.beams[0])


@pytest.mark.parametrize("spacing,want_big", [((0.5, 0.5), True), ((1.0, 1.0), False)])
def test_large_radii_go_through_the_second_sweep_launch(orc, engine, synth, spacing, want_big):
    """Rays 0.5 mm apart double the radii in pixels: batch radii up to the reference's limit of 32 (every level of
    k_superpose_sweep_big and its vector row |dy| = 32 run). k_superpose_sweep superposes the tiles of radius <= 16 and writes the slices,
    k_superpose_sweep_big adds the rest (launched until a finished compute has told the host that the field has none). Same field object
    computed three times: first with what the host launches blind, then with what the hint names; identical bits every time, and every
    intermediate against the oracle."""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=4, pitch=6.0, n_layers=3, angles=[0.0], steps=200, ct=ct)
    beam = scenarios.make_field(synth, 96, 256.0 / 96, (-128.0, -128.0, -106.0), 0.0, 4, 6.0, 3, 21, steps=200, ray_spacing=spacing, weight_lo=400.0)
    scn = scenarios.Scenario("rays %g mm" % spacing[0], synth, ct, scn.spacing, [beam])
    dose, ref, timing, info = _compare_field(orc, engine, scn, beam)
    assert (info["max_radius"] > 16) == want_big, info["max_radius"]
    if want_big:
        assert info["max_radius"] == 32
    n = scn.n_voxels
    with engine.Engine(0) as eng:
        eng.set_luts(synth)
        eng.set_ct(ct)
        d = eng.device_alloc(4 * n)
        fld = eng.create_field(beam, scn.dims)
        for _ in range(3):
            eng.device_zero(d, 4 * n)
            fld.compute(d)
            fld.finish()
            out = np.empty_like(ct)
            eng.to_host(out, d)
            np.testing.assert_array_equal(out, dose)
        fld.destroy()
        eng.device_free(d)


@pytest.mark.parametrize("case", ["radii <= 16", "radii up to 32"])
def test_the_two_general_superposition_kernels_agree(orc, engine, synth, monkeypatch, case):
    """The row sweep (k_superpose_sweep + k_superpose_sweep_big: source rows swept, T[|dy|][x] on the matrix cores) and k_superpose_mfma
    (output tiles visited) on the same heterogeneous field — 17 layers with radii within the first launch's reach, and 0.5 mm rays
    with radii up to 32, where the second launch adds the tiles of radius 17 .. 32: BEV doses within 1e-5 of each other relative to
    the maximum region (same weights up to the series threshold, different order of the sums), identical support, both within the parity
    bar of the oracle."""
    if case == "radii <= 16":
        ct, _ = scenarios.hetero_phantom(128)
        scn = scenarios.hetero_ct(synth, n=128, spots=7, pitch=6.0, n_layers=17, angles=[20.0], source_dist=(1900.0, 2300.0), ct=ct)
    else:
        ct, _ = scenarios.hetero_phantom(96)
        beam = scenarios.make_field(synth, 96, 256.0 / 96, (-128.0, -128.0, -106.0), 0.0, 5, 6.0, 4, 23, steps=200, ray_spacing=(0.5, 0.5), weight_lo=400.0)
        scn = scenarios.Scenario("rays 0.5 mm", synth, ct, (256.0 / 96,) * 3, [beam])
    ref = np.zeros_like(scn.ct)
    of = orc.run_field(scn, scn.beams[0], ref, keep_layers=True)
    W, H, L = of.info["ray_dims"]
    obev = of.get("bev").reshape(-1, H + 64, W + 64)
    res = {}
    for name, env in (("sweep", None), ("mfma", "1")):
        if env is None:
            monkeypatch.delenv("RTD_NO_SWEEP", raising=False)
        else:
            monkeypatch.setenv("RTD_NO_SWEEP", env)
        eng, fld, dose, timing, info, d_dose = _run_engine(engine, scn, scn.beams[0])
        assert (info["max_radius"] > 16) == (case != "radii <= 16")
        try:
            res[name] = (fld.fetch("bev").reshape(-1, H + 64, W + 64).copy(), dose.copy())
        finally:
            fld.destroy(); eng.device_free(d_dose); eng.close()
    monkeypatch.delenv("RTD_NO_SWEEP", raising=False)
    (bs, ds), (bm, dm) = res["sweep"], res["mfma"]
    big = obev > 1e-3 * obev.max()
    assert (np.abs(bs.astype(np.float64) - bm)[big] / obev[big]).max() <= 1e-5
    np.testing.assert_array_equal(bs[first_slice(of):] == 0, bm[first_slice(of):] == 0)
    for b in (bs, bm):
        _rel_close(b, obev, rtol=1e-4)
    _rel_close(ds, ref, rtol=1e-4)
    _rel_close(dm, ref, rtol=1e-4)


def first_slice(of):
    return int(of.info["beam_first_inside"])


def test_wide_field_many_tiles(orc, engine, synth):
    """A field wider than the CT: 448 x 448 rays -> 128 superposition output tiles (> the 64 that get a work-ranked dispatch
    order) and 1568 fill blocks (> 4 per CU: the plain longest-first placement); every intermediate still matches."""
    ct, _ = scenarios.hetero_phantom(64)
    scn = scenarios.hetero_ct(synth, n=64, spots=70, pitch=6.0, n_layers=2, angles=[0.0], steps=160, ct=ct)
    _, _, _, info = _compare_field(orc, engine, scn, scn.beams[0])
    assert info["ray_dims"][0] * info["ray_dims"][1] >= 400 * 400


def test_options_switches(orc, engine, synth):
    """DOSE_TO_WATER off, NO_NOZZLE, different cut-offs (CMakeLists.txt:36-79) follow the oracle too."""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=5, pitch=8.0, n_layers=2, angles=[20.0], ct=ct)
    opt = abi.default_options()
    opt.dose_to_water = 0
    opt.nozzle = 0
    opt.bp_depth_cutoff = 1.03
    opt.ks_sigma_cutoff = 2.5
    opt.conv_sigma_cutoff = 2.5
    opt.ray_weight_cutoff = 1.2
    _compare_field(orc, engine, scn, scn.beams[0], options=opt)


def test_reference_shaped_call_accumulates_two_beams(orc, engine, synth):
    """rtd_compute == cudaWrapperProtons semantics: incoming dose is kept and every beam is added (kernel_wrapper.cu:542,:92)."""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=5, pitch=8.0, n_layers=2, angles=[0.0, 90.0], ct=ct)
    base = np.full_like(scn.ct, 1e-7)
    ref = orc.compute(scn, dose=base.copy())
    dose = base.copy()
    import io
    log = io.StringIO()
    engine.cudaWrapperProtons(scn.ct, dose, scn.beams, scn.luts, log)
    assert "Total global execution time" in log.getvalue()
    _rel_close(dose, ref, rtol=1e-4)
    assert dose.min() >= 1e-7 * 0.999


def test_lut_directory_loader_matches_arrays(orc, engine, synth, tmp_path):
    """rtd_load_luts_dir (text layout of energy_reader.cpp) gives the same dose as rtd_set_luts with the parsed arrays."""
    from raytracedicom_amd import luts
    d = str(tmp_path / "luts")
    luts.write_lut_dir(d, synth)
    parsed = luts.read_lut_dir(d, water_cube_test=True)
    scn = scenarios.water_cube(parsed, n=64, n_layers=1, spots=9, pitch=5.0)
    doses = []
    for mode in ("dir", "arrays"):
        eng = engine.Engine(0)
        if mode == "dir":
            eng.load_luts_dir(d, True)
        else:
            eng.set_luts(parsed)
        eng.set_ct(scn.ct)
        dose = np.zeros_like(scn.ct)
        eng.compute(scn.beams, dose)
        doses.append(dose)
        eng.close()
    _rel_close(doses[0], doses[1], rtol=1e-5)
    assert doses[0].max() > 0


def test_error_paths(engine, synth):
    eng = engine.Engine(0)
    scn = scenarios.water_cube(synth, n=32, n_layers=1, spots=3)
    with pytest.raises(engine.RtdError) as e:
        eng.create_field(scn.beams[0], scn.dims)                 # LUTs / CT not set
    assert e.value.status == abi.RTD_ERR_NOT_READY
    with pytest.raises(engine.RtdError) as e:
        eng.load_luts_dir("/nonexistent/dir", False)
    assert e.value.status == abi.RTD_ERR_IO and "Failed to open" in str(e.value)
    eng.set_luts(synth)
    eng.set_ct(scn.ct)
    # radius overflow: a huge ray-pixel-to-sigma ratio (0.05 mm rays) needs radius > 32 -> reference throws (kernel_wrapper.cu:965)
    beam = scenarios.make_field(synth, 32, 8.0, (-128.0, -128.0, -106.0), 0.0, 3, 1.0, 1, 5, ray_spacing=(0.05, 0.05), steps=256, weight_lo=1e5)
    dose = np.zeros_like(scn.ct)
    with pytest.raises(engine.RtdError) as e:
        eng.compute([beam], dose)
    assert e.value.status == abi.RTD_ERR_RADIUS_OVERFLOW and "larger than allowed kernel superposition radius" in str(e.value)
    eng.close()


def test_cpp_shim_water_cube_driver(engine, synth, tmp_path):
    """The C++ host side (include/rtd_wrapper.hpp + examples/water_cube_main.cpp = the reference's main.cu water-cube
    branch) runs end to end through the C ABI and writes dose.dat like the reference (main.cu:211-216)."""
    import os
    import subprocess
    from conftest import ROOT
    from raytracedicom_amd import luts
    d = str(tmp_path / "luts")
    luts.write_lut_dir(d, synth)
    exe = str(tmp_path / "water_cube")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "examples"),
                           os.path.join(ROOT, "examples", "water_cube_main.cpp"),
                           "-L", os.path.join(ROOT, "raytracedicom_amd"), "-lrtd_hip", "-Wl,-rpath," + os.path.join(ROOT, "raytracedicom_amd"), "-o", exe])
    r = subprocess.run([exe, d + "/", str(tmp_path), "64", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Total global execution time" in r.stdout and "Max:" in r.stdout
    dose = np.fromfile(str(tmp_path / "dose.dat"), dtype=np.float32).reshape(64, 64, 64)
    assert dose.max() > 0 and float(r.stdout.strip().splitlines()[-1].split(":")[1]) == pytest.approx(float(dose.max()), rel=1e-5)
    prof = dose[:, 32, 32]
    depth = 128.0 - (4.0 * int(prof.argmax()) - 106.0)     # 4 mm voxels; beam starts at z = 128 mm
    assert 90.0 < depth < 125.0                               # Bragg peaks of the first two layers (~100-105 mm)
    # the reference's flag surface (examples/raytracedicom_main.cpp, config.cpp:13-51) drives the same plan: identical file
    cli = str(tmp_path / "raytracedicom")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "examples"),
                           os.path.join(ROOT, "examples", "raytracedicom_main.cpp"),
                           "-L", os.path.join(ROOT, "raytracedicom_amd"), "-lrtd_hip", "-Wl,-rpath," + os.path.join(ROOT, "raytracedicom_amd"), "-o", cli])
    out2 = tmp_path / "cli_out"
    out2.mkdir()
    cfg = tmp_path / "run.ini"
    cfg.write_text("output_directory = \"%s\"\nlut_dir = %s\nwater_cube = true\nwater_cube_edge = 32\nlayers = 2\n" % (out2, d))
    r2 = subprocess.run([cli, "--config_file", str(cfg), "--water_cube_edge", "64", "--gpu_id", "0"], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr
    assert "water_cube_edge=64" in r2.stdout and "Max:" in r2.stdout
    dose2 = np.fromfile(str(out2 / "dose.dat"), dtype=np.float32)
    assert np.array_equal(dose2, dose.ravel())


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8, 13, 21, 34])
def test_seeded_random_scenarios(orc, engine, synth, seed):
    """Seeded sweep over the knobs that select code paths (layer count vs layer groups, spot grid / ray grid shape, rotation,
    divergence, step count, sharp and broad spots): every intermediate and the dose against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([48, 64, 96]))
    ct, _ = scenarios.hetero_phantom(n, seed=int(rng.integers(1, 99)))
    spots = int(rng.integers(1, 13))
    pitch = float(rng.choice([2.5, 4.0, 6.0, 9.0]))
    n_layers = int(rng.choice([1, 2, 7, 14, 15, 17, 23]))
    deg = float(rng.choice([0.0, 13.0, 90.0, 141.0, 270.0]))
    dist = (math.inf, math.inf) if rng.random() < 0.4 else (float(rng.uniform(900, 3000)), float(rng.uniform(900, 3000)))
    steps = int(rng.choice([97, 200, 256, 333]))
    scn = scenarios.hetero_ct(synth, n=n, spots=spots, pitch=pitch, n_layers=n_layers, angles=[deg], source_dist=dist, steps=steps, ct=ct)
    _compare_field(orc, engine, scn, scn.beams[0])


@pytest.mark.parametrize("n_samples,n_hu", [(6144, 3072), (2048, 9000)])
def test_lookup_tables_too_large_for_lds(orc, engine, n_samples, n_hu):
    """Tables beyond the LDS staging limits take the global-memory paths: k_fill<false> (two cIDD rows + 1/X0 above 48 KiB) and the
    tail loop of the tracer's LUT staging (density + SP beyond the 4096 entries held in registers)."""
    from raytracedicom_amd import luts
    big = luts.synth_luts(n_energies=40, n_samples=n_samples, n_hu=n_hu)
    ct, _ = scenarios.hetero_phantom(64)
    scn = scenarios.hetero_ct(big, n=64, spots=5, pitch=7.0, n_layers=3, angles=[15.0], ct=ct)
    _compare_field(orc, engine, scn, scn.beams[0])


def test_c1_bev_dose_against_the_reference_cpu_convolution(orc, engine, synth):
    """BASELINE.json configs[0] (water cube 128^3, one G000 field, one energy layer) against the REFERENCE'S OWN CPU code: in water
    a BEV slice has one sigma, so its superposition is xConvCpuScat + yConvCpu of the slice (src/cpu_convolution_1d.cpp, compiled from
    the reference's sources into oracle/_ref/libref.so, which travels to the GPU box as a built file; the oracle's restatement of the
    two routines, bit-identical to them by golden vector G7, stands in where it is absent). The HIP engine's BEV dose — per-voxel-sigma
    MFMA superposition, Taylor-series weights — must agree to 2e-5 relative above 1e-3 of the maximum (same weights to ~1e-7, different
    summation order)."""
    from oracle import ref_cpu_path
    scn = scenarios.water_cube(synth, n=128, n_layers=1)
    ref_dose = np.zeros_like(scn.ct)
    of = orc.run_field(scn, scn.beams[0], ref_dose, keep_layers=True)
    which = "reference" if ref_cpu_path.ref_lib() is not None else "port"
    sep = ref_cpu_path.separable_bev(of, scn.beams[0], which=which)
    assert sep is not None
    sep_bev, _, n_slices, max_rad = sep
    eng, fld, dose, timing, info, d_dose = _run_engine(engine, scn, scn.beams[0])
    try:
        W, H, L = info["ray_dims"]
        gbev = fld.fetch("bev").reshape(-1, H + 64, W + 64)
        sb = sep_bev[:gbev.shape[0]]
        assert n_slices > 50 and sb.max() > 0
        big = sb > 1e-3 * sb.max()
        rel = np.abs(gbev.astype(np.float64) - sb)[big] / sb[big]
        assert info["uniform_sigma"] == 1                            # water: the separable superposition kernel took the field
        print("HIP BEV vs %s CPU convolution: %d slices, radius <= %d, max rel diff %.3g" % (which, n_slices, max_rad, rel.max()))
        assert rel.max() <= 2e-5
        assert np.abs(gbev.astype(np.float64) - sb).max() <= 2e-6 * sb.max()
    finally:
        fld.destroy()
        eng.device_free(d_dose)
        eng.close()


@pytest.mark.parametrize("case", ["rows staged in LDS", "wave per row block", "wide ray grid: two strips", "fine rays: radii above 16",
                                  "forty layers: more than the weight table holds"])
def test_uniform_sigma_kernels(orc, engine, synth, monkeypatch, case):
    """The two separable-superposition kernels of rtd_uniform.hpp on water fields, every intermediate against the oracle:
    k_superpose_uniform4 (a block per four row blocks of a slice, the layers' rows within reach staged in LDS: ray grids of up to 128
    columns; on 0.6 mm rays the deep layers' radii pass 16 and their rows are read from global memory by the same kernel, through its
    generic instruction pattern; with forty layers the weights of the layers past the block's table of 32 are computed per wave),
    k_superpose_uniform2 (one wave per row block and strip of 192 columns, no staging: forced with
    RTD_UNIFORM_V2 on the same field, and chosen by the engine for a ray grid of 192 columns, where a slice has two strips)."""
    if case == "wave per row block":
        monkeypatch.setenv("RTD_UNIFORM_V2", "1")
    else:
        monkeypatch.delenv("RTD_UNIFORM_V2", raising=False)
    if case == "wide ray grid: two strips":
        scn = scenarios.water_cube(synth, n=128, n_layers=3, spots=47, pitch=3.0)
    elif case == "fine rays: radii above 16":
        scn = scenarios.water_cube(synth, n=128, n_layers=4, spots=12, pitch=3.0, ray_spacing=(0.6, 0.6))
    elif case == "forty layers: more than the weight table holds":
        scn = scenarios.water_cube(synth, n=64, n_layers=40, spots=9, pitch=3.0)      # (the shallow slices are crossed by all 40: kU4TabLayers = 32)
    else:
        scn = scenarios.water_cube(synth, n=128, n_layers=4)
    _, _, _, info = _compare_field(orc, engine, scn, scn.beams[0])
    assert info["uniform_sigma"] == 1
    if case == "wide ray grid: two strips":
        assert info["ray_dims"][0] > 128
    if case == "fine rays: radii above 16":
        assert info["ray_dims"][0] <= 128 and info["max_radius"] > 16, info


def test_uniform_sigma_path_equals_the_general_superposition(orc, engine, synth, monkeypatch):
    """A water field is superposed by the separable kernels of rtd_uniform.hpp (one sigma per slice), decided on the device. With
    the path disabled (RTD_NO_UNIFORM_PATH, read at field creation) the same field goes through the row sweep: both BEV doses agree
    to rounding (same weights, different order of the sums), both are within the parity tolerance of the oracle, and the dose too.
    A beam that leaves the water (air gap in the CT) is NOT uniform and must take the general path by itself."""
    scn = scenarios.water_cube(synth, n=128, n_layers=3)
    ref = np.zeros_like(scn.ct)
    of = orc.run_field(scn, scn.beams[0], ref, keep_layers=True)
    W, H, L = of.info["ray_dims"]
    obev = of.get("bev").reshape(-1, H + 64, W + 64)
    res = {}
    for name, env in (("uniform", None), ("general", "1")):
        if env is None:
            monkeypatch.delenv("RTD_NO_UNIFORM_PATH", raising=False)
        else:
            monkeypatch.setenv("RTD_NO_UNIFORM_PATH", env)
        eng, fld, dose, timing, info, d_dose = _run_engine(engine, scn, scn.beams[0])
        try:
            assert info["uniform_sigma"] == (1 if env is None else 0)
            res[name] = (fld.fetch("bev").reshape(-1, H + 64, W + 64).copy(), dose.copy())
        finally:
            fld.destroy(); eng.device_free(d_dose); eng.close()
    monkeypatch.delenv("RTD_NO_UNIFORM_PATH", raising=False)
    bu, du = res["uniform"]; bg, dg = res["general"]
    big = obev > 1e-3 * obev.max()
    assert (np.abs(bu.astype(np.float64) - bg)[big] / obev[big]).max() <= 1e-5
    np.testing.assert_array_equal(bu == 0, bg == 0)                  # same support: same radii
    for b in (bu, bg):
        _rel_close(b, obev, rtol=1e-4)
    _rel_close(du, ref, rtol=1e-4)
    _rel_close(dg, ref, rtol=1e-4)
    # half of the water replaced by a density step across the field: slices are no longer uniform
    ct2 = scn.ct.copy()
    ct2[:, :, : ct2.shape[2] // 2] *= 1.3
    scn2 = scenarios.Scenario("water with a density step", scn.luts, ct2, scn.spacing, scn.beams)
    dose2, ref2, timing, info = _compare_field(orc, engine, scn2, scn2.beams[0])
    assert info["uniform_sigma"] == 0
    # a divergent beam into the same water: the rays' step lengths differ, so do their sigmas — general path, same parity bar
    scn3 = scenarios.water_cube(synth, n=128, n_layers=3, source_dist=(1800.0, 2200.0))
    dose3, ref3, timing, info = _compare_field(orc, engine, scn3, scn3.beams[0])
    assert info["uniform_sigma"] == 0
    # One field object across computes: what it learned about its input (uniform: the general kernel is not even launched the next
    # time; heterogeneous: no detection, no separable launch) must not outlive that input — a new CT on the handle resets it.
    n = scn.n_voxels
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        d = eng.device_alloc(4 * n)
        fld = eng.create_field(scn.beams[0], scn.dims)
        outs = []
        for ct_now, want_uniform in ((None, 1), (None, 1), (ct2, 0), (ct2, 0), (scn.ct, 1), (scn.ct, 1)):
            if ct_now is not None:
                eng.set_ct(ct_now)
            eng.device_zero(d, 4 * n)
            fld.compute(d)
            _, info = fld.finish()
            assert info["uniform_sigma"] == want_uniform
            h = np.empty_like(scn.ct)
            eng.to_host(h, d)
            outs.append(h)
        np.testing.assert_array_equal(outs[0], du)
        np.testing.assert_array_equal(outs[1], du)
        np.testing.assert_array_equal(outs[2], dose2)
        np.testing.assert_array_equal(outs[3], dose2)
        np.testing.assert_array_equal(outs[4], du)
        np.testing.assert_array_equal(outs[5], du)
        fld.destroy()
        eng.device_free(d)
