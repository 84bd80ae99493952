"""RT ion plan spot list -> BeamSettings (include/rtd_plan.hpp), the step the reference's main.cu stops
before (main.cu:150-197). CPU: the C++ unit checks of tests/cpp/test_rtd_plan.cpp. GPU: the reference's water-cube field
written as a spot list and read back through the CLI gives the dose of the built-in plan."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from raytracedicom_amd import engine


def _cxx(src, exe, extra=()):
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "examples"),
                           src, "-L", os.path.join(ROOT, "raytracedicom_amd"), "-lrtd_hip", "-Wl,-rpath," + os.path.join(ROOT, "raytracedicom_amd"),
                           "-o", exe, *extra])


def test_spot_list_to_beam_settings(tmp_path):
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    exe = str(tmp_path / "test_rtd_plan")
    _cxx(os.path.join(ROOT, "tests", "cpp", "test_rtd_plan.cpp"), exe)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "lut_small") + "/", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "rtd_plan ok" in r.stdout, r.stderr


def _against_oracle(orc, synth, dump, dose_file, voxel, water=True):
    """The CLI's dose.dat against the oracle run on exactly what the CLI passed through the C ABI (RTD_DUMP_CALL)."""
    import call_dump
    from raytracedicom_amd import luts
    scn = call_dump.read(dump, luts.read_lut_dir(os.path.dirname(dump) + "/luts", water_cube_test=water), spacing=(voxel,) * 3)
    ref = orc.compute(scn)
    dose = np.fromfile(dose_file, dtype=np.float32).reshape(ref.shape)
    mx = float(ref.max())
    assert mx > 0
    thr = ref > 1e-3 * mx
    assert (np.abs(dose - ref)[thr] <= 1e-4 * ref[thr] + 1e-6 * mx).all(), float((np.abs(dose - ref)[thr] / ref[thr]).max())
    assert np.abs(dose - ref).max() <= 1e-5 * mx
    rate, n_eval, _ = orc.gamma_pass_rate(ref, dose, scn.spacing)
    assert rate == 1.0 and n_eval > 0
    return scn


@pytest.mark.gpu
def test_cli_spot_list_reproduces_the_builtin_plan(orc, synth, tmp_path):
    from raytracedicom_amd import luts
    d = str(tmp_path / "luts")
    luts.write_lut_dir(d, synth)
    cli = str(tmp_path / "raytracedicom")
    _cxx(os.path.join(ROOT, "examples", "raytracedicom_main.cpp"), cli)
    outs = []
    for name in ("builtin", "spots"):
        o = tmp_path / name
        o.mkdir()
        outs.append(o)
    base = [cli, "--water_cube", "--water_cube_edge", "64", "--layers", "3", "--lut_dir", d]
    env = dict(os.environ, RTD_DUMP_CALL=str(tmp_path / "builtin.call"))
    r = subprocess.run(base + ["--output_directory", str(outs[0]), "--fine_grained_timing"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    # the reference's FINE_GRAINED_TIMING text (kernel_wrapper.cu:598,604,1298-1307,1325,1349-1352)
    for line in ("Copy data to GPU and bind to textures:", "Calculating field no. 0", "Time to trace 128x128 rays 512 steps:",
                 "Time depositing IDD and calculating sigma 3 time(s):", "Time executing superposition 3 time(s):", "Kernel time to transform",
                 "Time to copy dose back to host:", "Approximate total execution time (excluding GPU initialisation):"):
        assert line in r.stdout, line
    # f1 against the oracle, not against itself: the water-cube CLI run
    _against_oracle(orc, synth, str(tmp_path / "builtin.call"), str(outs[0] / "dose.dat"), 256.0 / 64)
    spots = str(tmp_path / "spots.txt")
    r = subprocess.run(base + ["--output_directory", str(outs[1]), "--dump_spot_list", spots], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and os.path.exists(spots), r.stderr
    env = dict(os.environ, RTD_DUMP_CALL=str(tmp_path / "spots.call"))
    r = subprocess.run(base + ["--output_directory", str(outs[1]), "--spot_list", spots], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert "3267 spots in 3 layer(s) on a 33x33 spot grid" in r.stdout
    assert "Total global execution time (excluding GPU initialisation):" in r.stdout
    # f2 against the oracle: the spot-list run
    _against_oracle(orc, synth, str(tmp_path / "spots.call"), str(outs[1] / "dose.dat"), 256.0 / 64)
    a = np.fromfile(str(outs[0] / "dose.dat"), dtype=np.float32)
    b = np.fromfile(str(outs[1] / "dose.dat"), dtype=np.float32)
    assert a.max() > 0
    # sigma -> FWHM -> sigma costs an ulp; everything else is identical
    np.testing.assert_allclose(b, a, rtol=2e-5, atol=1e-7 * float(a.max()))
