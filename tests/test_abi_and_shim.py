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
    assert engine.lib().rtd_abi_version() == 1


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
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "water_cube_main.cpp"), "-L", os.path.join(ROOT, "raytracedicom_amd"),
                           "-lrtd_hip", "-Wl,-rpath," + os.path.join(ROOT, "raytracedicom_amd"), "-o", exe])
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "lut_small") + "/", str(tmp_path), "16", "1"],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


def test_product_path_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "raytracedicom_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f in (), "%s mentions the oracle" % os.path.join(dirpath, f)
    for f in ("rtd.h", "rtd_wrapper.hpp", "rtd_types.hpp"):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().lower()
