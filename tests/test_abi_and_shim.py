"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol include/rtd.h declares,
fails loudly without a GPU (no CPU fallback), and the C++ shim + example driver compile against it."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT
from raytracedicom_amd import abi, engine


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rtd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtd_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    lib = ctypes.CDLL(engine.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "librtd_hip.so does not export %s" % n
    assert engine.lib().rtd_abi_version() == abi.RTD_ABI_VERSION == 3
    import re
    assert int(re.search(r"#define RTD_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "rtd.h")).read()).group(1)) == abi.RTD_ABI_VERSION


def test_struct_sizes_match_header():
    """The ctypes mirrors must have the C layout: compile a tiny C program printing sizeof/offsetof."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "rtd.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rtd_affine), sizeof(rtd_idx_transform), sizeof(rtd_beam), sizeof(rtd_luts),
 sizeof(rtd_options), sizeof(rtd_timing), sizeof(rtd_field_info), offsetof(rtd_beam, gantry_to_dose_idx));return 0;}
'''
    exe = os.path.join(ROOT, "tests", "_abi_probe")
    subprocess.run(["gcc", "-x", "c", "-I", os.path.join(ROOT, "include"), "-o", exe, "-"], input=src.encode(), check=True)
    try:
        out = subprocess.check_output([exe]).decode().split()
    finally:
        os.remove(exe)
    got = [ctypes.sizeof(abi.RtdAffine), ctypes.sizeof(abi.RtdIdxTransform), ctypes.sizeof(abi.RtdBeam), ctypes.sizeof(abi.RtdLuts),
           ctypes.sizeof(abi.RtdOptions), ctypes.sizeof(abi.RtdTiming), ctypes.sizeof(abi.RtdFieldInfo),
           abi.RtdBeam.gantry_to_dose_idx.offset]
    assert got == [int(x) for x in out]


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(engine.RtdError) as e:
        engine.Engine(0)
    assert e.value.status == abi.RTD_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_cpp_shim_and_example_compile(tmp_path):
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    exe = str(tmp_path / "water_cube")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "examples"),
                           os.path.join(ROOT, "examples", "water_cube_main.cpp"), "-L", os.path.join(ROOT, "raytracedicom_amd"),
                           "-lrtd_hip", "-Wl,-rpath," + os.path.join(ROOT, "raytracedicom_amd"), "-o", exe])
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "lut_small") + "/", str(tmp_path), "16", "1"],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


def _build_cli(tmp_path):
    exe = str(tmp_path / "raytracedicom")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "examples"),
                           os.path.join(ROOT, "examples", "raytracedicom_main.cpp"), "-L", os.path.join(ROOT, "raytracedicom_amd"),
                           "-lrtd_hip", "-Wl,-rpath," + os.path.join(ROOT, "raytracedicom_amd"), "-o", exe])
    return exe


def test_cli_flag_surface_of_the_reference(tmp_path):
    """examples/raytracedicom_main.cpp: the reference's flags (config.cpp:13-51) — required arguments, existing file / directory
    checks, --config_file overridden by the command line, the echoed configuration; DICOM input is parsed before the engine is needed."""
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    exe = _build_cli(tmp_path)
    luts = os.path.join(ROOT, "tests", "golden", "lut_small")
    out = str(tmp_path / "out"); os.mkdir(out)
    run = lambda *a: subprocess.run([exe, *a], capture_output=True, text=True)
    assert run("--help").returncode == 0 and "--output_directory" in run("--help").stdout
    r = run("--water_cube", "--lut_dir", luts)
    assert r.returncode == 2 and "--output_directory is required" in r.stderr
    r = run("--water_cube", "--lut_dir", luts, "--output_directory", str(tmp_path / "missing"))
    assert r.returncode == 2 and "Directory does not exist" in r.stderr
    r = run("--output_directory", out, "--lut_dir", luts)
    assert r.returncode == 2 and "--ct_dir is required" in r.stderr
    r = run("--output_directory", out, "--lut_dir", luts, "--ct_dir", out, "--rtplan", str(tmp_path / "nope.dcm"), "--beams", "G000")
    assert r.returncode == 2 and "File does not exist" in r.stderr
    r = run("--output_directory", out, "--bogus", "1")
    assert r.returncode == 2 and "not expected" in r.stderr
    r = run("--output_directory", out, "--gpu_id", "-1", "--water_cube")
    assert r.returncode == 2 and "could not convert" in r.stderr
    # DICOM mode: CT series + RT Ion Plan are parsed (include/rtd_dicom.hpp) before the engine is created; two beams are accepted
    # (the reference throws "Multi-beam calculation not yet supported", main.cu:117-120)
    import numpy as np
    import dicom_fixture as dfx
    ct_dir = str(tmp_path / "ct")
    dfx.write_ct_series(ct_dir, np.zeros((6, 8, 8), np.int32), (4.0, 4.0, 4.0), (-16.0, -16.0, -12.0))
    plan = str(tmp_path / "plan.dcm")
    lay = dict(energy=100.0, fwhm=(10.0, 10.0), x=np.array([-3.0, 0.0, 3.0]), y=np.zeros(3), w=np.ones(3, "f4"))
    dfx.write_ion_plan(plan, [dict(name="G000", gantry=0.0, iso=(0, 0, 0), vsad=(2000.0, 2000.0), layers=[lay]),
                              dict(name="G090", gantry=90.0, iso=(0, 0, 0), vsad=(2000.0, 2000.0), layers=[lay])])
    cfg = str(tmp_path / "run.ini")
    open(cfg, "w").write("# run parameters\noutput_directory = \"%s\"\nlut_dir=%s\nct_dir = %s ; series\nrtplan = %s\nbeams = [\"G000\", \"G090\"]\ngpu_id=3\n"
                         % (out, luts, ct_dir, plan))
    import torch
    gpu = torch.cuda.is_available()
    r = run("--config_file", cfg, "--gpu_id", "0")                                            # the command line overrides the file
    assert 'beams=["G000", "G090"]' in r.stdout and "gpu_id=0" in r.stdout
    assert "Loading field 1 corresponding to beamname G090" in r.stdout and "3 spots in 1 layer(s) on a 3x1 spot grid" in r.stdout
    if not gpu:
        assert r.returncode == 1 and "no CPU fallback" in r.stderr
    r = run("--config_file", cfg, "--beams", "G045")
    assert r.returncode == 1 and "no beam named G045" in r.stderr and 'beams=["G045"]' in r.stdout
    r = run("--config_file", cfg, "--ct_dir", str(tmp_path / "out"))
    assert r.returncode == 1 and "contains no DICOM Series" in r.stderr
    if not gpu:
        r = run("--config_file", cfg, "--water_cube", "--water_cube_edge", "16", "--layers", "1", "--gpu_id", "0")
        assert r.returncode == 1 and "no CPU fallback" in r.stderr and "water_cube=true" in r.stdout


def test_graft_entry_build():
    """__graft_entry__.build() is what the driver runs on the CPU box every round."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    importlib.import_module("__graft_entry__").build()


def test_product_path_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "raytracedicom_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f in (), "%s mentions the oracle" % os.path.join(dirpath, f)
    for f in sorted(os.listdir(os.path.join(ROOT, "include"))):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().lower()
