"""GPU tests of size-independent properties and edge cases (through the C ABI):
bitwise reproducibility (the engine uses no float atomics), exact linearity in the spot weights, the reference water
cube C2 at full size, beams that never enter the patient, ragged sizes (steps / layers / spot maps that are not
multiples of any tile or batch size)."""
import math

import numpy as np
import pytest

from raytracedicom_amd import abi, scenarios

pytestmark = pytest.mark.gpu


def _dose(engine, scn, beams=None, options=None):
    dose = np.zeros_like(scn.ct)
    with engine.Engine(0) as eng:
        if options is not None:
            eng.set_options(options)
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        eng.compute(beams if beams is not None else scn.beams, dose)
    return dose


def test_bitwise_reproducible(engine, synth):
    """Two runs of the same plan give identical bits (the reference's atomicAdd flush does not)."""
    ct, _ = scenarios.hetero_phantom(128)
    scn = scenarios.hetero_ct(synth, n=128, spots=6, pitch=7.0, n_layers=5, angles=[30.0], ct=ct)
    a = _dose(engine, scn)
    b = _dose(engine, scn)
    assert a.max() > 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_exact_linearity_in_spot_weights(engine, synth):
    """Scaling every spot weight by 4 (exact in binary) scales every stage, hence the dose, by exactly 4 when the
    ray-weight cut-off cannot change the live set (RAY_WEIGHT_CUTOFF = 0). (A power of FOUR: the row sweep folds the square root of a
    source's dose into both factors of its patch, (sqrt(d) e_i)(sqrt(d) e_j) — exactly scalable only when the root of the factor is a
    power of two.)"""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=5, pitch=8.0, n_layers=3, angles=[0.0], ct=ct)
    opt = abi.default_options()
    opt.ray_weight_cutoff = 0.0
    d1 = _dose(engine, scn, options=opt)
    b = scn.beams[0]
    b2 = scenarios.BeamSettings(b.spotWeights * np.float32(4.0), b.beamEnergies, b.spotSigmas, b.raySpacing, b.tracerSteps, b.sourceDist,
                                b.spotIdxToGantry, b.gantryToImIdx, b.gantryToDoseIdx)
    d2 = _dose(engine, scn, beams=[b2], options=opt)
    assert d1.max() > 0
    np.testing.assert_array_equal(d2, d1 * np.float32(4.0))


def test_c2_reference_water_cube_full_size(orc, engine, synth):
    """BASELINE.json configs[1]: the reference's WATER_CUBE_TEST, 256^3, 33x33 spots x 20 layers, 512 steps (a water field: the engine
    takes it through k_superpose_uniform4, the separable superposition on the matrix cores — tests/test_gpu_parity.py checks the path)."""
    scn = scenarios.water_cube(synth, n=256, n_layers=20)
    ref = orc.compute(scn)
    dose = _dose(engine, scn)
    mx = float(ref.max())
    thr = ref > 0.1 * mx
    rel = np.abs(dose - ref)[thr] / ref[thr]
    assert rel.max() < 1e-4, rel.max()
    assert np.abs(dose - ref).max() < 1e-5 * mx
    rate, n_eval, gmax = orc.gamma_pass_rate(ref, dose, scn.spacing)
    assert rate == 1.0 and n_eval > 100000
    # spread-out Bragg peak: 20 layers -> a plateau between the first and the last peak depth on the central axis
    prof = ref[:, 128, 128]
    z = np.nonzero(prof > 0.5 * prof.max())[0]
    assert (z.max() - z.min()) > 60


def test_beam_that_never_enters_the_patient(orc, engine, synth):
    """All-air CT: nothing is inside, the field is skipped (the reference would launch with negative sizes)."""
    scn = scenarios.water_cube(synth, n=48, n_layers=2, spots=5, pitch=6.0)
    scn.ct[:] = 0.0
    dose = _dose(engine, scn)
    assert dose.max() == 0.0 and dose.min() == 0.0
    ref = orc.compute(scn)
    assert ref.max() == 0.0


@pytest.mark.parametrize("steps,n_layers,spots", [(300, 23, 3), (77, 1, 1), (513, 11, 7)])
def test_ragged_sizes(orc, engine, synth, steps, n_layers, spots):
    """Steps, layers and spot maps that are not multiples of the 32-step trace segments, the 16/8-step batches or the
    10 layer groups; a single spot; more layers than groups."""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=spots, pitch=9.0, n_layers=n_layers, angles=[10.0], steps=steps, ct=ct)
    ref = orc.compute(scn)
    dose = _dose(engine, scn)
    mx = float(ref.max())
    if mx > 0:
        assert np.abs(dose - ref).max() <= 2e-5 * mx
        thr = ref > 0.1 * mx
        assert (np.abs(dose - ref)[thr] / ref[thr]).max() < 1e-4
    else:
        assert dose.max() == 0.0


def test_fetch_unknown_name_and_double_finish(engine, synth):
    scn = scenarios.water_cube(synth, n=48, n_layers=1, spots=5, pitch=6.0)
    eng = engine.Engine(0)
    eng.set_luts(scn.luts)
    eng.set_ct(scn.ct)
    d = eng.device_alloc(4 * scn.n_voxels)
    eng.device_zero(d, 4 * scn.n_voxels)
    fld = eng.create_field(scn.beams[0], scn.dims)
    with pytest.raises(engine.RtdError):
        fld.finish()                                  # not computed yet
    fld.compute(d)
    t1, i1 = fld.finish()
    t2, i2 = fld.finish()
    assert i1 == i2 and t1["total_ms"] > 0
    with pytest.raises(engine.RtdError) as e:
        fld.fetch("no_such_buffer")
    assert "unknown name" in str(e.value)
    fld.destroy()
    eng.device_free(d)
    eng.close()


def test_clear_dose_resets_exactly_the_written_box(engine, synth):
    """rtd_field_clear_dose zeroes every voxel the field's compute changed (so a zero volume is zero again) and nothing
    outside the device-side dose box (a volume pre-filled with ones keeps its ones there)."""
    ct, _ = scenarios.hetero_phantom(96)
    scn = scenarios.hetero_ct(synth, n=96, spots=4, pitch=7.0, n_layers=3, angles=[20.0], ct=ct)
    n = scn.n_voxels
    eng = engine.Engine(0)
    eng.set_luts(scn.luts)
    eng.set_ct(scn.ct)
    d = eng.device_alloc(4 * n)
    fld = eng.create_field(scn.beams[0], scn.dims)
    with pytest.raises(engine.RtdError):
        fld.clear_dose(d)                             # nothing computed yet
    eng.device_zero(d, 4 * n)
    fld.compute(d)
    fld.finish()
    out = np.empty(n, dtype=np.float32)
    eng.to_host(out, d)
    assert out.max() > 0
    fld.clear_dose(d)
    eng.sync()
    eng.to_host(out, d)
    assert not out.any()
    ones = np.ones(n, dtype=np.float32)
    eng.to_device(d, ones)
    fld.compute(d)
    _, info = fld.finish()
    eng.to_host(out, d)
    changed = out != 1.0
    assert changed.any()
    fld.clear_dose(d)
    eng.sync()
    eng.to_host(out, d)
    assert (out[changed] == 0.0).all()                # everything the compute touched is cleared
    vol = out.reshape(scn.ct.shape)
    lo, hi = info["bbox_min"], info["bbox_max"]
    outside = np.ones(vol.shape, dtype=bool)
    outside[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = False
    assert (vol[outside] == 1.0).all()                # the clear stays inside the reported bounding box
    fld.destroy()
    eng.device_free(d)
    eng.close()


def test_field_launch_sequence_is_hipgraph_capturable(engine, synth):
    """rtd_field_clear_dose + rtd_field_compute make no host round trip, allocation or synchronisation, so the whole plan
    iteration can be captured into a hipGraph on the caller's stream; replays reproduce the directly launched dose bit for bit."""
    import torch
    n = 96
    ct, _ = scenarios.hetero_phantom(n)
    scn = scenarios.hetero_ct(synth, n=n, spots=5, pitch=7.0, n_layers=4, angles=[25.0], ct=ct)
    dev = torch.device("cuda:0")
    eng = engine.Engine(0)
    eng.set_luts(scn.luts)
    ct_dev = torch.from_numpy(scn.ct).to(dev)
    eng.set_ct_device(ct_dev.data_ptr(), scn.dims)
    fld = eng.create_field(scn.beams[0], scn.dims)
    s = torch.cuda.Stream()
    eng.set_stream(s.cuda_stream)
    ref = torch.zeros((n, n, n), dtype=torch.float32, device=dev)
    dose = torch.zeros_like(ref)
    with torch.cuda.stream(s):
        fld.compute(ref.data_ptr())
        fld.finish()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fld.clear_dose(dose.data_ptr())
            fld.compute(dose.data_ptr())
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    assert float(ref.max()) > 0 and torch.equal(dose, ref)
    eng.set_stream(None)
    fld.destroy()
    eng.close()


def test_reduction_tree_hand_offs_are_race_free_over_many_launches(engine, synth):
    """The superposition adds the layer groups' accumulators in a tree whose hand-offs cross XCDs (agent-scope stores / loads, node
    counters that the second arriver resets). 40 launches of the bench-sized field (14 groups, 15 output tiles, ~230 live slices:
    ~40 k hand-offs per launch) must give the same BEV dose bit for bit, and the counters must be back at zero every time (a stale
    slot or a lost arrival would change bits or leave a tile unwritten)."""
    import zlib
    ct, _ = scenarios.hetero_phantom(256)
    scn = scenarios.hetero_ct(synth, n=256, angles=[0.0], ct=ct)      # same ray grid, layers and steps as the 512^3 bench field
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        f = eng.create_field(scn.beams[0], scn.dims)
        first = None
        for it in range(40):
            f.compute_bev()
            f.finish()
            bev = f.fetch("bev")
            h = zlib.crc32(bev.tobytes())
            if first is None:
                first, ref = h, bev.copy()
                assert ref.max() > 0
            elif h != first:
                bad = np.nonzero(bev.view(np.uint32) != ref.view(np.uint32))[0]
                raise AssertionError("launch %d differs from launch 0 in %d BEV values (first at flat index %d)" % (it, bad.size, int(bad[0])))
        f.destroy()
