"""GPU tests of the pieces a multi-GPU plan is made of (all through the C ABI, on ONE GPU): the two halves of a field,
the exported BEV slab and the clipped transfers, and the in-process plan (rtd_plan_*) driven with several handles on the
same device. Bit-identity with the sequential single-GPU call is the bar: every voxel receives the same `+=` sequence."""
import math

import numpy as np
import pytest

from raytracedicom_amd import abi, scenarios

pytestmark = pytest.mark.gpu


def _scn(synth, n=128, angles=(0.0, 90.0, 37.0), dist=(math.inf, math.inf)):
    ct, _ = scenarios.hetero_phantom(n)
    return scenarios.hetero_ct(synth, n=n, spots=6, pitch=7.0, n_layers=4, angles=list(angles), source_dist=dist, ct=ct)


def _sequential(engine, scn, base=None):
    dose = np.zeros_like(scn.ct) if base is None else base.copy()
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        eng.compute(scn.beams, dose)
    return dose


def test_bev_then_transfer_equals_compute(engine, synth):
    """rtd_field_compute == rtd_field_compute_bev + rtd_field_transfer; clipped transfers into disjoint boxes add up to it."""
    scn = _scn(synth, angles=(30.0,))
    n = scn.n_voxels
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        f = eng.create_field(scn.beams[0], scn.dims)
        d = [eng.device_alloc(4 * n) for _ in range(3)]
        for p in d:
            eng.device_zero(p, 4 * n)
        f.compute(d[0])
        f.finish()
        f.compute_bev()
        f.transfer(d[1])
        nz = scn.dims[2]
        cuts = [0, nz // 3, nz // 2, nz]
        for a, b in zip(cuts[:-1], cuts[1:]):
            f.transfer(d[2], (0, 0, a), (scn.dims[0] - 1, scn.dims[1] - 1, b - 1))
        t, info = f.finish()
        out = [np.empty_like(scn.ct) for _ in range(3)]
        for o, p in zip(out, d):
            eng.to_host(o, p)
        assert out[0].max() > 0 and t["transforming_ms"] >= 0
        np.testing.assert_array_equal(out[1], out[0])
        np.testing.assert_array_equal(out[2], out[0])
        # clear of a clipped box: zero inside, untouched outside
        f.clear_dose_box(d[0], (0, 0, 0), (scn.dims[0] - 1, scn.dims[1] - 1, nz // 2 - 1))
        eng.to_host(out[1], d[0])
        assert out[1][:nz // 2].max() == 0.0
        np.testing.assert_array_equal(out[1][nz // 2:], out[0][nz // 2:])
        f.destroy()
        for p in d:
            eng.device_free(p)


@pytest.mark.parametrize("deg", [0.0, 90.0, 52.0])
def test_transfer_init_writes_the_box(engine, synth, deg):
    """rtd_field_transfer_init into a volume holding stale values inside the field's dose box (and zeros elsewhere) == a transfer
    into a zeroed volume, bit for bit (plain and transposed kernels)."""
    scn = _scn(synth, angles=(deg,), dist=(1900.0, 2100.0))
    n = scn.n_voxels
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        f = eng.create_field(scn.beams[0], scn.dims)
        a, b = eng.device_alloc(4 * n), eng.device_alloc(4 * n)
        eng.device_zero(a, 4 * n)
        eng.device_zero(b, 4 * n)
        f.compute(a)
        _, info = f.finish()
        want = np.empty_like(scn.ct)
        eng.to_host(want, a)
        lo, hi = info["dose_box_min"], info["dose_box_max"]
        stale = np.zeros_like(scn.ct)
        stale[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = 7.5            # garbage inside the box only
        eng.to_device(b, stale)
        f.compute_bev()
        f.transfer_init(b)
        f.finish()
        got = np.empty_like(scn.ct)
        eng.to_host(got, b)
        assert want.max() > 0
        np.testing.assert_array_equal(got, want)
        f.destroy()
        eng.device_free(a); eng.device_free(b)


@pytest.mark.parametrize("deg,dist", [(0.0, (math.inf, math.inf)), (90.0, (1800.0, 2200.0)), (141.0, (math.inf, math.inf))])
def test_exported_bev_slab_transfers_identically_on_another_handle(engine, synth, deg, dist):
    """The message [state record | non-zero block of the BEV dose] attached to a geometry-only field on ANOTHER handle gives the
    same dose bits as the field's own transfer (plain and transposed transfer kernels, divergent beam)."""
    scn = _scn(synth, angles=(deg,), dist=dist)
    n = scn.n_voxels
    a, b = engine.Engine(0), engine.Engine(0)
    try:
        a.set_luts(scn.luts)
        a.set_ct(scn.ct)
        f = a.create_field(scn.beams[0], scn.dims)
        da = a.device_alloc(4 * n)
        a.device_zero(da, 4 * n)
        f.compute(da)
        info, nbytes = f.wait_plan()
        assert 2048 < nbytes <= f.message_bound()
        msg = a.device_alloc(nbytes)
        f.export_bev(msg, nbytes)
        f.finish()
        a.sync()
        # (how the slab compares with the dose box it turns into depends on voxel size against ray spacing: 2 mm voxels here
        #  make the box the smaller one; on the 0.5 mm grid of the bench workload the slab is 6-8 times smaller)
        r = b.create_field(scn.beams[0], scn.dims, remote=True)         # needs neither LUTs nor CT on handle b
        db = b.device_alloc(4 * n)
        b.device_zero(db, 4 * n)
        r.attach_bev(msg)
        r.transfer(db)
        _, rinfo = r.finish()
        assert rinfo["dose_box_min"] == info["dose_box_min"] and rinfo["dose_box_max"] == info["dose_box_max"]
        x, y = np.empty_like(scn.ct), np.empty_like(scn.ct)
        a.to_host(x, da)
        b.to_host(y, db)
        assert x.max() > 0
        np.testing.assert_array_equal(y, x)
        # a message buffer that is too small is reported, and the receiver's volume stays untouched
        small = a.device_alloc(4096)
        f.export_bev(small, 4096)
        a.sync()
        b.device_zero(db, 4 * n)
        r.attach_bev(small)
        r.transfer(db)
        with pytest.raises(engine.RtdError):
            r.finish()
        b.to_host(y, db)
        assert y.max() == 0.0
        r.destroy(); f.destroy()
        for e, p in ((a, da), (a, msg), (a, small), (b, db)):
            e.device_free(p)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("dist", [(math.inf, math.inf), (1800.0, 2200.0)])
def test_fused_transfer_of_several_fields_equals_the_sequence(engine, synth, dist):
    """rtd_fields_transfer_init: own fields and attached slabs at angles that take all three gather layouts (lanes along x, y, z)
    written into a box in ONE launch == the loop of rtd_field_transfer over the same fields into a zeroed volume, bit for bit;
    stale values inside the box are overwritten, voxels outside the box are not touched; the box need not be brick-aligned."""
    angles = (0.0, 90.0, 37.0, 180.0, 270.0, 141.0)
    scn = _scn(synth, angles=angles, dist=dist)
    n = scn.n_voxels
    nx, ny, nz = scn.dims
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        own = [eng.create_field(b, scn.dims) for b in scn.beams]
        want_d, got_d = eng.device_alloc(4 * n), eng.device_alloc(4 * n)
        eng.device_zero(want_d, 4 * n)
        msgs, fields = [], []
        for i, f in enumerate(own):
            f.compute_bev()
            f.transfer(want_d)                                        # the sequence: field 0, 1, 2, ... accumulated
            if i % 2:                                                 # every other field goes through the message path
                info, nbytes = f.wait_plan()
                m = eng.device_alloc(nbytes)
                f.export_bev(m, nbytes)
                r = eng.create_field(scn.beams[i], scn.dims, remote=True)
                r.attach_bev(m)
                msgs.append(m); fields.append(r)
            else:
                fields.append(f)
            f.finish()
        want = np.empty_like(scn.ct)
        eng.to_host(want, want_d)
        assert want.max() > 0
        # (a) the whole grid
        stale = np.full_like(scn.ct, 3.25)
        eng.to_device(got_d, stale)
        eng.transfer_fields_init(fields, got_d)
        t, _ = own[4].finish()                                        # the last own field of the list carries the launch's events
        assert t["total_ms"] > 0
        got = np.empty_like(scn.ct)
        eng.to_host(got, got_d)
        np.testing.assert_array_equal(got, want)
        # (b) an unaligned box: inside = the sequence, outside untouched
        lo, hi = (5, 9, 33), (nx - 7, ny - 20, 77)
        eng.to_device(got_d, stale)
        eng.transfer_fields_init(fields, got_d, lo, hi)
        own[4].finish()
        eng.to_host(got, got_d)
        inside = np.zeros(scn.ct.shape, dtype=bool)
        inside[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = True
        np.testing.assert_array_equal(got[inside], want[inside])
        assert (got[~inside] == 3.25).all()
        # (c) slabs along z written one by one tile the volume
        eng.to_device(got_d, stale)
        cuts = [0, 40, 41, 90, nz]
        for a, b in zip(cuts[:-1], cuts[1:]):
            eng.transfer_fields_init(fields, got_d, (0, 0, a), (nx - 1, ny - 1, b - 1))
        own[4].finish()
        eng.to_host(got, got_d)
        np.testing.assert_array_equal(got, want)
        # more than 16 fields in one call is refused
        with pytest.raises(engine.RtdError):
            eng.transfer_fields_init(fields * 3, got_d)
        for f in fields + own:
            f.destroy()
        for p in msgs + [want_d, got_d]:
            eng.device_free(p)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_in_process_plan_equals_sequential_call(orc, engine, synth, devices):
    """rtd_plan_compute with 1, 2 and 3 handles (threads) on one GPU: bit-identical to rtd_compute on one handle — incoming dose
    kept, beams dealt round-robin, BEV slabs exchanged, every handle transferring all beams into its z-slab — and equal to the
    oracle within the dose tolerance."""
    scn = _scn(synth, angles=(0.0, 90.0, 37.0, 180.0))
    base = np.full_like(scn.ct, 1e-7)
    want = _sequential(engine, scn, base)
    dose = base.copy()
    with engine.Plan(devices) as plan:
        plan.set_luts(scn.luts)
        plan.set_ct(scn.ct)
        per_beam, pt = plan.compute(scn.beams, dose)
        assert pt["n_devices"] == len(devices) and pt["total_ms"] > 0 and len(per_beam) == 4
        np.testing.assert_array_equal(dose, want)
        # second call on the same plan object (workspaces and message buffers are reused)
        dose2 = base.copy()
        plan.compute(scn.beams, dose2)
        np.testing.assert_array_equal(dose2, want)
    ref = orc.compute(scn, dose=base.copy())
    mx = float(ref.max())
    assert np.abs(dose - ref).max() <= 1e-4 * mx
    rate, n_eval, _ = orc.gamma_pass_rate(ref, dose, scn.spacing)
    assert rate == 1.0 and n_eval > 0


def test_in_process_plan_reports_device_side_errors_and_keeps_the_volume(engine, synth):
    """A beam whose superposition radius overflows fails the whole plan call (kernel_wrapper.cu:965) and the caller's dose
    buffer is left as it was."""
    scn = scenarios.water_cube(synth, n=32, n_layers=1, spots=3)
    good = scn.beams[0]
    bad = scenarios.make_field(synth, 32, 8.0, (-128.0, -128.0, -106.0), 0.0, 3, 1.0, 1, 5, ray_spacing=(0.05, 0.05), steps=256, weight_lo=1e5)
    dose = np.full_like(scn.ct, 0.25)
    with engine.Plan([0, 0]) as plan:
        plan.set_luts(scn.luts)
        plan.set_ct(scn.ct)
        with pytest.raises(engine.RtdError) as e:
            plan.compute([good, bad], dose)
        assert e.value.status == abi.RTD_ERR_RADIUS_OVERFLOW
    assert (dose == 0.25).all()


def test_radius_overflow_leaves_the_device_volume_untouched(engine, synth):
    """The split form: on a radius overflow neither the superposition nor the transfer runs, so dev_dose keeps its contents
    (the reference throws before any superposition launch, kernel_wrapper.cu:965)."""
    scn = scenarios.water_cube(synth, n=32, n_layers=1, spots=3)
    bad = scenarios.make_field(synth, 32, 8.0, (-128.0, -128.0, -106.0), 0.0, 3, 1.0, 1, 5, ray_spacing=(0.05, 0.05), steps=256, weight_lo=1e5)
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        n = scn.n_voxels
        d = eng.device_alloc(4 * n)
        init = np.full_like(scn.ct, 3.0)
        eng.to_device(d, init)
        f = eng.create_field(bad, scn.dims)
        f.compute(d)
        with pytest.raises(engine.RtdError) as e:
            f.finish()
        assert e.value.status == abi.RTD_ERR_RADIUS_OVERFLOW
        out = np.empty_like(scn.ct)
        eng.to_host(out, d)
        np.testing.assert_array_equal(out, init)
        f.destroy()
        eng.device_free(d)


def test_released_workspace_is_taken_over(engine, synth):
    """rtd_field_release + rtd_field_create of the same shape reuses the device workspace; results are those of fresh fields."""
    scn = _scn(synth, angles=(0.0, 180.0))
    want = _sequential(engine, scn)
    n = scn.n_voxels
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        d = eng.device_alloc(4 * n)
        eng.device_zero(d, 4 * n)
        for beam in scn.beams:
            f = eng.create_field(beam, scn.dims)
            f.compute(d)
            f.finish()
            f.release()
        out = np.empty_like(scn.ct)
        eng.to_host(out, d)
        np.testing.assert_array_equal(out, want)
        eng.device_free(d)


def test_released_workspace_is_not_taken_over_by_a_transposed_ray_grid(engine, synth):
    """Two beams with the same number of rays, steps, layers and spots but transposed ray grids (64 x 32 against 32 x 64) need
    different workspaces: the padded BEV cube is (W + 64) x (H + 64) x S and the superposition's hand-off slots / node counters
    scale with ceil(bevW / 64) * ceil(bevH / 32) (6 against 8 output tiles here). A released workspace must only go to a field of
    the same W AND H; the result of the second beam equals a fresh engine's bit for bit."""
    ct, _ = scenarios.hetero_phantom(64)

    def beam_of(spots, pitch):
        return scenarios.hetero_ct(synth, n=64, spots=spots, pitch=pitch, n_layers=2, angles=[0.0], steps=160, ct=ct, seed=5).beams[0]
    # 4 x 1 spots at 9 mm -> 64 x 32 rays; 2 x 2 spots at (1, 27) mm -> 32 x 64 rays: same ray count, spot count, intermediate size
    a, b = beam_of((4, 1), 9.0), beam_of((2, 2), (1.0, 27.0))
    scn = scenarios.hetero_ct(synth, n=64, spots=2, pitch=4.0, n_layers=2, angles=[0.0], steps=160, ct=ct)
    n = scn.n_voxels
    want = _sequential(engine, scenarios.Scenario("b", synth, ct, scn.spacing, [b]))
    with engine.Engine(0) as eng:
        eng.set_luts(synth)
        eng.set_ct(ct)
        d = eng.device_alloc(4 * n)
        fa = eng.create_field(a, scn.dims)
        eng.device_zero(d, 4 * n)
        fa.compute(d)
        _, ia = fa.finish()
        fa.release()
        fb = eng.create_field(b, scn.dims)
        eng.device_zero(d, 4 * n)
        fb.compute(d)
        _, ib = fb.finish()
        assert ia["ray_dims"][:2] == [64, 32] and ib["ray_dims"][:2] == [32, 64], (ia["ray_dims"], ib["ray_dims"])
        out = np.empty_like(ct)
        eng.to_host(out, d)
        np.testing.assert_array_equal(out, want)
        fb.release()
        eng.device_free(d)


def test_uniform_hint_belongs_to_the_inputs_of_the_launch(orc, engine, synth):
    """compute (water) -> set_ct (heterogeneous) -> finish -> compute: what the first compute learned (one sigma per slice) is
    recorded for the CT it was LAUNCHED under, not for the one bound when the host got round to finishing it; the second compute
    must run the general superposition and give the heterogeneous field's dose."""
    scn = scenarios.water_cube(synth, n=96, n_layers=2, spots=9, pitch=5.0)
    ct2 = scn.ct.copy()
    ct2[:, :, : ct2.shape[2] // 2] *= 1.3
    scn2 = scenarios.Scenario("water with a density step", scn.luts, ct2, scn.spacing, scn.beams)
    want2 = _sequential(engine, scn2)
    n = scn.n_voxels
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts)
        eng.set_ct(scn.ct)
        d = eng.device_alloc(4 * n)
        fld = eng.create_field(scn.beams[0], scn.dims)
        eng.device_zero(d, 4 * n)
        fld.compute(d)
        _, info = fld.finish()
        assert info["uniform_sigma"] == 1                            # the hint is now "uniform" under the water CT
        eng.device_zero(d, 4 * n)
        fld.compute(d)                                               # launched as a known-uniform field
        eng.set_ct(ct2)                                              # ... the CT changes before the host finishes it
        _, info = fld.finish()
        assert info["uniform_sigma"] == 1
        eng.device_zero(d, 4 * n)
        fld.compute(d)
        _, info = fld.finish()
        assert info["uniform_sigma"] == 0
        out = np.empty_like(scn.ct)
        eng.to_host(out, d)
        np.testing.assert_array_equal(out, want2)
        fld.destroy()
        eng.device_free(d)


def test_in_process_plan_over_rccl_transport(orc, engine, synth, monkeypatch):
    """RTD_PLAN_TRANSPORT=rccl: the slabs travel by ncclBroadcast on communicators made with ncclCommInitAll (librccl opened on demand).
    One GPU allows one rank only; RTD_PLAN_SELF_MESSAGES makes the owner transfer from the broadcast copy, so export -> RCCL ->
    attach -> transfer all run: same dose bits as the sequential call. Duplicate device ids are refused (one RCCL rank per GPU)."""
    scn = _scn(synth, angles=(0.0, 90.0))
    want = _sequential(engine, scn)
    monkeypatch.setenv("RTD_PLAN_TRANSPORT", "rccl")
    monkeypatch.setenv("RTD_PLAN_SELF_MESSAGES", "1")
    dose = np.zeros_like(scn.ct)
    with engine.Plan([0]) as plan:
        plan.set_luts(scn.luts)
        plan.set_ct(scn.ct)
        _, pt = plan.compute(scn.beams, dose)
        assert pt["exchange_ms"] > 0
    np.testing.assert_array_equal(dose, want)
    with pytest.raises(engine.RtdError):
        engine.Plan([0, 0])


def test_bench_exchange_path_under_rccl_with_one_rank():
    """bench.py --exchange-selftest: the N>1 code path (nccl process group, all-gathers, packed message, fused slab transfer,
    pipelining on two volumes) with a world of one rank — what can be run of it on a one-GPU box. The volume it leaves equals the
    plain transfer of the field bit for bit (reduce_check 0) and the reuse of a dirty volume is clean (clear_check)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--exchange-selftest", "--size", "128", "--steps", "4", "--warmup", "2",
                        "--no-cpu"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 1 and r["clear_check"] is True and r["reduce_check_rel_err"] == 0.0
    assert "all-gather" in r["config"]["exchange"] and r["value"] > 0


@pytest.mark.parametrize("cut", ["measured", "rank 1 gets an empty slab"])
def test_bench_launches_its_own_ranks_two_processes_one_gpu(cut):
    """(Second case: the slab cut is told that rank 1's field is hopelessly late, so its slab holds nothing that any field's dose box reaches (or is empty) — it
    transfers nothing and attaches nothing; the eight-field plan has such a rank — and the run must still produce its line, checks included.)
    `python bench.py --gpus 2` with no launcher in front: bench.py starts torch.distributed.run as a child process before it
    touches the GPU and relays rank 0's line. Two ranks share this box's one GPU over gloo (the RCCL world needs one GPU per rank):
    the N>1 path with two real ranks — two slabs gathered, each rank writing its slab with both fields. The slabs together hold the
    sum of the fields (reduce_check), a reused volume is clean (clear_check), and the volume assembled on rank 0 equals the
    sequential beam loop of the reference (kernel_wrapper.cu:601, :92) bit for bit (assembled_check)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "128", "--steps", "4",
                        "--warmup", "2", "--no-cpu"] + ([] if cut == "measured" else ["--debug-head-us", "600,90000"]),
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["scaling"] == "weak"
    assert r["reduce_check_rel_err"] < 1e-9 and r["clear_check"] is True and r["assembled_check"] is True
    assert r["value"] > 0 and "all-gather" in r["config"]["exchange"]
    if cut != "measured":       # rank 1 is left with the rows no field's dose box reaches (or none): nothing to transfer there
        import re
        m = re.search(r"ranges \[\((\d+), (\d+)\), \((\d+), (-?\d+)\)\]", r["config"]["exchange"])
        assert m and int(m.group(4)) - int(m.group(3)) + 1 <= 16, r["config"]["exchange"]


@pytest.mark.parametrize("angles", [(0.0,), (90.0, 37.0, 200.0)])
def test_deferred_ct_uploads_only_what_the_rays_cross(engine, synth, angles):
    """rtd_set_ct_deferred / rtd_plan_set_ct_deferred: every field uploads the index box of the CT its tracer can sample. The device
    volume is poisoned first (a NaN-filled CT of the same size is bound eagerly, so the reused allocation holds NaN wherever no box
    lands): the dose must still equal the eager path bit for bit — i.e. no sample ever falls outside the uploaded boxes — for
    parallel and divergent, axis-aligned and oblique beams; the in-process plan moves only the changed block of the dose volume
    (incoming dose kept, inside and outside that block)."""
    for dist in ((math.inf, math.inf), (1500.0, 1900.0)):
        scn = _scn(synth, n=128, angles=angles, dist=dist)
        base = np.random.default_rng(3).random(scn.ct.shape, dtype=np.float32)
        want = _sequential(engine, scn, base=base)
        poison = np.full_like(scn.ct, np.nan)
        with engine.Engine(0) as eng:
            eng.set_luts(scn.luts)
            eng.set_ct(poison)
            eng.set_ct(scn.ct, deferred=True)
            got = base.copy()
            eng.compute(scn.beams, got)
        np.testing.assert_array_equal(got, want)
        with engine.Plan([0, 0]) as pl:
            pl.set_luts(scn.luts)
            pl.set_ct(poison)
            pl.set_ct(scn.ct, deferred=True)
            got = base.copy()
            pl.compute(scn.beams, got)
        np.testing.assert_array_equal(got, want)


def test_bench_line_keeps_the_contract():
    """bench.py (small CT, few steps): ONE JSON line with every key of the driver's contract, the roofline and cpu_baseline objects, the
    self-checks true and the parity leg green."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "128", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r["roofline"], k
    assert abs(r["roofline"]["frac"] - r["roofline"]["achieved"] / r["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in r["cpu_baseline"], k
    assert r["cpu_baseline"]["kind"] in ("port", "reference") and r["cpu_baseline"]["value"] > 0
    assert r["clear_check"] is True and r["parity"]["gamma_1pct_1mm_pass"] == 1.0 and r["parity"]["max_rel_diff_above_10pct"] < 1e-4
    assert r["ms_plan_end_to_end"]["matches_resident_path"] is True
