"""DICOM input without ITK/GDCM (include/rtd_dicom.hpp). The reader is checked against what the fixture
writer (tests/dicom_fixture.py) put into the files — parity with ITK itself is unpinned (ITK is not available here)."""
import os
import subprocess

import numpy as np
import pytest

import dicom_fixture as dfx
from conftest import ROOT


def _build(tmp_path):
    exe = str(tmp_path / "test_rtd_dicom")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_rtd_dicom.cpp"), "-o", exe])
    return exe


def _plan():
    rng = np.random.default_rng(5)
    layers = []
    for e in (101.5, 108.25, 120.0):
        xs, ys = np.meshgrid(np.arange(-10.0, 10.1, 5.0), np.arange(-7.5, 7.6, 2.5))
        keep = rng.random(xs.size) > 0.2                              # ragged layers
        layers.append(dict(energy=e, fwhm=(9.5 + 0.01 * e, 10.25), x=xs.ravel()[keep], y=ys.ravel()[keep], w=(1 + rng.random(int(keep.sum()))).astype("f4")))
    return [dict(name="G000", gantry=0.0, couch=0.0, collimator=0.0, iso=(1.0, 2.0, 3.0), vsad=(2000.0, 2560.0), layers=layers),
            dict(name="G270", gantry=270.0, couch=10.0, collimator=5.0, iso=(-4.0, 0.5, 12.0), vsad=(1800.0, 1800.0), layers=layers[:2])]


@pytest.mark.parametrize("syntax,undefined", [(dfx.EXPLICIT, True), (dfx.EXPLICIT, False), (dfx.IMPLICIT, True), (dfx.IMPLICIT, False)])
def test_reader_returns_what_the_writer_wrote(tmp_path, syntax, undefined):
    exe = _build(tmp_path)
    rng = np.random.default_rng(11)
    hu = rng.integers(-1000, 2000, size=(7, 12, 10)).astype(np.int32)
    spacing, origin = (0.9765625, 1.25, 2.5), (-120.5, -88.25, 40.0)
    ct_dir = str(tmp_path / "ct")
    dfx.write_ct_series(ct_dir, hu, spacing, origin, slope=1.0, intercept=-1024.0, syntax=syntax)
    # a second series with a later UID in the same directory is ignored (the reference takes the first series UID)
    dfx.write_ct_series(ct_dir, hu[:2] * 0, spacing, origin, syntax=syntax, series_uid="1.2.826.0.1.3680043.8.498.9", shuffle=False, prefix="AA")
    plan = str(tmp_path / "plan.dcm")
    beams = _plan()
    dfx.write_ion_plan(plan, beams, syntax=syntax, undefined_length=undefined)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([exe, ct_dir, plan, "G270", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "beams: G000 G270" in r.stdout
    ct = np.fromfile(str(out / "ct.bin"), dtype=np.float32)
    geo = np.fromfile(str(out / "ct_geo.bin"), dtype=np.float32)
    assert geo[:3].tolist() == [10, 12, 7]
    np.testing.assert_array_equal(ct.reshape(7, 12, 10), (hu + 1000).astype(np.float32))           # HU + 1000, slices re-sorted
    np.testing.assert_allclose(geo[3:12].reshape(3, 3), np.diag(spacing), rtol=1e-6)                 # Direction * diag(Spacing)
    np.testing.assert_allclose(geo[12:15], origin, rtol=1e-6)
    spots = np.fromfile(str(out / "spots.bin"), dtype=np.float32).reshape(-1, 6)
    b = beams[1]
    exp = np.concatenate([np.stack([np.full(len(l["x"]), l["energy"]), l["x"], l["y"], np.full(len(l["x"]), l["fwhm"][0]),
                                    np.full(len(l["x"]), l["fwhm"][1]), l["w"]], axis=1) for l in b["layers"]]).astype(np.float32)
    np.testing.assert_array_equal(spots, exp)
    bg = np.fromfile(str(out / "beam_geo.bin"), dtype=np.float32)
    np.testing.assert_allclose(bg, [270.0, 10.0, 5.0, -4.0, 0.5, 12.0, 1800.0, 1800.0, 2.0])
    # IEC 61217 / head-first-supine conventions of gantryToPatientHfs: rows = (gantry, support) angle pairs,
    # columns = images of the source point (0,0,1000), of X_g and of Y_g in DICOM patient coordinates (x left, y posterior, z superior)
    conv = np.fromfile(str(out / "conventions.bin"), dtype=np.float32).reshape(5, 3, 3)
    np.testing.assert_allclose(conv[0], [[0, -1000, 0], [1, 0, 0], [0, 0, 1]], atol=1e-4)      # gantry 0: source anterior, X_g = left, Y_g = superior
    np.testing.assert_allclose(conv[1][0], [1000, 0, 0], atol=1e-3)                            # gantry 90: source at the patient's left
    np.testing.assert_allclose(conv[2][0], [0, 1000, 0], atol=1e-3)                            # gantry 180: source posterior
    np.testing.assert_allclose(conv[3][0], [-1000, 0, 0], atol=1e-3)                           # gantry 270: source at the patient's right
    np.testing.assert_allclose(conv[4][0], [0, -1000, 0], atol=1e-3)                           # a table rotation leaves a vertical beam vertical
    np.testing.assert_allclose(np.abs(conv[4][1]), [0, 0, 1], atol=1e-4)                       # ... and turns X_g into the patient's long axis
    # tracer range covers the volume along the beam axis of G270 (support angle 10 deg): project the 8 volume corners on Z_g
    sd, st = np.fromfile(str(out / "tracer_range.bin"), dtype=np.float32)
    g, t = np.radians(270.0), -np.radians(10.0)
    ry = np.array([[np.cos(g), 0, np.sin(g)], [0, 1, 0], [-np.sin(g), 0, np.cos(g)]])
    rz = np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]])
    m = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0.0]]) @ rz @ ry                               # gantry -> patient
    corners = np.array([[origin[a] + spacing[a] * ((hu.shape[2 - a] - 0.5) if (c >> a) & 1 else -0.5) for a in range(3)] for c in range(8)])
    zg = (np.linalg.inv(m) @ (corners - np.array([-4.0, 0.5, 12.0])).T)[2]
    assert sd >= zg.max() and sd - zg.max() < 2.5                                              # step 0 just upstream of the volume
    assert st >= sd - zg.min() and st - (sd - zg.min()) < 3.0                                  # and enough 1 mm steps to leave it


def test_oblique_series_rescale_and_errors(tmp_path):
    exe = _build(tmp_path)
    rng = np.random.default_rng(3)
    hu = (rng.integers(-500, 500, size=(4, 6, 8)) * 2).astype(np.int32)
    c, s = np.cos(0.3), np.sin(0.3)
    orient = (c, s, 0.0, -s, c, 0.0)                                  # rotated about z
    ct_dir = str(tmp_path / "ct")
    dfx.write_ct_series(ct_dir, hu, (1.0, 1.5, 3.0), (10.0, -20.0, 5.0), orientation=orient, slope=2.0, intercept=-2000.0)
    plan = str(tmp_path / "plan.dcm")
    dfx.write_ion_plan(plan, _plan())
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([exe, ct_dir, plan, "G000", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ct = np.fromfile(str(out / "ct.bin"), dtype=np.float32).reshape(4, 6, 8)
    np.testing.assert_array_equal(ct, (hu + 1000).astype(np.float32))                               # slope 2, intercept -2000
    geo = np.fromfile(str(out / "ct_geo.bin"), dtype=np.float32)
    m = geo[3:12].reshape(3, 3)
    np.testing.assert_allclose(m[:, 0], np.array([c, s, 0.0]) * 1.0, atol=1e-6)                     # column 0 = row direction * dx
    np.testing.assert_allclose(m[:, 1], np.array([-s, c, 0.0]) * 1.5, atol=1e-6)
    np.testing.assert_allclose(m[:, 2], np.array([0.0, 0.0, 1.0]) * 3.0, atol=1e-6)
    # errors: unknown beam, a plan that is not a plan, an empty directory
    r = subprocess.run([exe, ct_dir, plan, "G123", str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "no beam named G123" in r.stderr
    r = subprocess.run([exe, ct_dir, os.path.join(ct_dir, "README.txt"), "G000", str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "not a DICOM Part 10 file" in r.stderr
    empty = tmp_path / "empty"
    empty.mkdir()
    r = subprocess.run([exe, str(empty), plan, "G000", str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "contains no DICOM Series" in r.stderr


@pytest.mark.gpu
def test_cli_dicom_input_reproduces_the_water_cube_plan(orc, synth, tmp_path):
    """End to end: a DICOM CT (water, oblique-free but with the slice normal along -y of the patient so that a gantry-0 beam of a
    head-first supine patient runs along the image's -k axis) + an RT Ion Plan holding the reference's water-cube field give,
    through rtd_dicom.hpp + rtd_plan.hpp + the C++ shim, the dose of the built-in WATER_CUBE_TEST plan; two beams add up."""
    from raytracedicom_amd import luts
    d = str(tmp_path / "luts")
    luts.write_lut_dir(d, synth)
    cli = str(tmp_path / "raytracedicom")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "examples"),
                           os.path.join(ROOT, "examples", "raytracedicom_main.cpp"), "-L", os.path.join(ROOT, "raytracedicom_amd"), "-lrtd_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "raytracedicom_amd"), "-o", cli])
    outs = {}
    for name in ("builtin", "dicom", "two"):
        outs[name] = tmp_path / name
        outs[name].mkdir()
    n, layers = 64, 2
    base = [cli, "--lut_dir", d]
    r = subprocess.run(base + ["--water_cube", "--water_cube_edge", str(n), "--layers", str(layers), "--output_directory", str(outs["builtin"])],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    spots_txt = str(tmp_path / "spots.txt")
    r = subprocess.run(base + ["--water_cube", "--layers", str(layers), "--output_directory", str(outs["builtin"]), "--dump_spot_list", spots_txt],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = np.array([[float(t) for t in ln.split()] for ln in open(spots_txt) if ln[0] not in "#gis"], dtype=np.float32)
    lay = []
    for e in np.unique(rows[:, 0]):
        m = rows[:, 0] == e
        lay.append(dict(energy=float(e), fwhm=(float(rows[m][0, 3]), float(rows[m][0, 4])), x=rows[m][:, 1], y=rows[m][:, 2], w=rows[m][:, 5]))
    voxel = 256.0 / n
    ct_dir = str(tmp_path / "ct")
    # image i -> patient x, image j -> patient z, slice normal = -y: voxel (i, j, k) sits at gantry (vi - 128, vj - 128, vk - 106) for gantry 0
    dfx.write_ct_series(ct_dir, np.zeros((n, n, n), np.int32), (voxel, voxel, voxel), (-128.0, 106.0, -128.0), orientation=(1, 0, 0, 0, 0, 1))
    plan = str(tmp_path / "plan.dcm")
    beam = dict(name="G000", gantry=0.0, iso=(0.0, 0.0, 0.0), vsad=(np.inf, np.inf), layers=lay)
    dfx.write_ion_plan(plan, [beam, dict(beam, name="AGAIN")])
    common = base + ["--ct_dir", ct_dir, "--rtplan", plan, "--start_depth", "128", "--tracer_steps", "512"]
    env = dict(os.environ, RTD_DUMP_CALL=str(tmp_path / "dicom.call"))
    r = subprocess.run(common + ["--beams", "G000", "--output_directory", str(outs["dicom"])], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    # f3 against the oracle, not against the built-in plan only: the CT and the beam the DICOM reader produced, as they crossed the C ABI
    from test_plan_import import _against_oracle
    scn = _against_oracle(orc, synth, str(tmp_path / "dicom.call"), str(outs["dicom"] / "dose.dat"), voxel, water=False)
    assert scn.ct.shape == (n, n, n) and float(scn.ct.min()) == 1000.0 == float(scn.ct.max())      # HU 0 -> HU+1000
    assert "%d spots in %d layer(s) on a 33x33 spot grid" % (33 * 33 * layers, layers) in r.stdout
    a = np.fromfile(str(outs["builtin"] / "dose.dat"), dtype=np.float32)
    b = np.fromfile(str(outs["dicom"] / "dose.dat"), dtype=np.float32)
    assert a.max() > 0
    np.testing.assert_allclose(b, a, rtol=2e-5, atol=1e-7 * float(a.max()))
    env2 = dict(os.environ, RTD_DUMP_CALL=str(tmp_path / "two.call"))
    r = subprocess.run(common + ["--beams", "G000", "AGAIN", "--output_directory", str(outs["two"]), "--gpu_ids", "0", "0"], capture_output=True,
                       text=True, timeout=300, env=env2)
    assert r.returncode == 0, r.stderr
    _against_oracle(orc, synth, str(tmp_path / "two.call"), str(outs["two"] / "dose.dat"), voxel, water=False)   # two beams on two handles (threads)
    c = np.fromfile(str(outs["two"] / "dose.dat"), dtype=np.float32)
    np.testing.assert_allclose(c, 2.0 * b, rtol=1e-6, atol=1e-7 * float(a.max()))
    # without overrides the tracer range is fitted to the CT: still the same physics (rays start outside the cube instead of inside it)
    r = subprocess.run(base + ["--ct_dir", ct_dir, "--rtplan", plan, "--beams", "G000", "--output_directory", str(outs["two"])],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    e = np.fromfile(str(outs["two"] / "dose.dat"), dtype=np.float32).reshape(n, n, n)
    prof, ref = e[:, n // 2, n // 2], a.reshape(n, n, n)[:, n // 2, n // 2]
    assert abs(int(prof.argmax()) - int(ref.argmax())) <= 6          # 22 mm more water upstream moves the peak by that much
