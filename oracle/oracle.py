"""ctypes binding of oracle/liboracle.so — the CPU oracle. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracedicom_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_find_max.restype = C.c_float
        L.orc_find_decimal_ordered.restype = C.c_float
        L.orc_vector_interpolate.restype = C.c_float
        L.orc_sample1d.restype = C.c_float
        L.orc_sample2d.restype = C.c_float
        L.orc_sample3d.restype = C.c_float
        L.orc_field_get.restype = C.c_void_p
        L.orc_field_error.restype = C.c_char_p
        L.orc_gamma_pass_rate.restype = C.c_double
        L.orc_field_run.argtypes = [C.POINTER(abi.RtdLuts), abi.c_float_p, C.POINTER(C.c_uint32), C.POINTER(abi.RtdBeam),
                                    abi.c_float_p, C.POINTER(C.c_uint32), C.POINTER(abi.RtdOptions), C.c_int,
                                    C.POINTER(C.c_void_p)]
        L.orc_compute.argtypes = [C.POINTER(abi.RtdLuts), abi.c_float_p, C.POINTER(C.c_uint32), C.POINTER(abi.RtdBeam),
                                  C.c_int, abi.c_float_p, C.POINTER(C.c_uint32), C.POINTER(abi.RtdOptions)]
        L.orc_field_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t)]
        L.orc_field_info.argtypes = [C.c_void_p, C.POINTER(abi.RtdFieldInfo)]
        L.orc_field_free.argtypes = [C.c_void_p]
        L.orc_field_error.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


_DTYPES = {"first_inside": np.int32, "first_outside": np.int32, "first_passive": np.int32, "eff_radius": np.int32,
           "tile_radius": np.uint8}


class OracleField:
    """Intermediates of one field, same names as rtd_field_fetch (include/rtd.h)."""

    def __init__(self, handle, status):
        self._h = handle
        self.status = status
        info = abi.RtdFieldInfo()
        lib().orc_field_info(handle, C.byref(info))
        self.info = info.as_dict()
        self.error = lib().orc_field_error(handle).decode()

    def get(self, name):
        n = C.c_size_t(0)
        p = lib().orc_field_get(self._h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        dt = np.dtype(_DTYPES.get(name, np.float32))
        buf = (C.c_char * n.value).from_address(p)
        return np.frombuffer(buf, dtype=dt).copy()

    def close(self):
        if self._h:
            lib().orc_field_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def run_field(scn, beam, dose, options=None, keep_layers=True):
    """orc_field_run: one field of scenario scn accumulated into dose ([Z][Y][X] float32, modified in place)."""
    opt = options or abi.default_options()
    la = scn.luts.as_abi()
    ba = beam.as_abi()
    h = C.c_void_p()
    st = lib().orc_field_run(C.byref(la), abi.fptr(scn.ct), abi.uint3(scn.dims), C.byref(ba), abi.fptr(dose),
                             abi.uint3(scn.dims), C.byref(opt), 1 if keep_layers else 0, C.byref(h))
    return OracleField(h, st)


def compute(scn, dose=None, options=None):
    """orc_compute: all beams, reference-shaped (accumulates into dose)."""
    opt = options or abi.default_options()
    if dose is None:
        dose = np.zeros_like(scn.ct)
    la = scn.luts.as_abi()
    ba = __import__("raytracedicom_amd.scenarios", fromlist=["beams_abi"]).beams_abi(scn.beams)
    st = lib().orc_compute(C.byref(la), abi.fptr(scn.ct), abi.uint3(scn.dims), ba, len(scn.beams), abi.fptr(dose),
                           abi.uint3(scn.dims), C.byref(opt))
    if st != 0:
        raise RuntimeError("oracle status %d" % st)
    return dose


def gamma_pass_rate(ref, ev, spacing, dd=0.01, dta=1.0, threshold=0.10):
    """Global gamma(dd, dta mm) pass rate of ev against ref above threshold*max(ref)."""
    ref = abi.f32(ref)
    ev = abi.f32(ev)
    dims = abi.uint3((ref.shape[2], ref.shape[1], ref.shape[0]))
    sp = (C.c_float * 3)(*[float(s) for s in spacing])
    n = C.c_int64(0)
    g = C.c_float(0)
    r = lib().orc_gamma_pass_rate(abi.fptr(ref), abi.fptr(ev), dims, sp, C.c_float(dd), C.c_float(dta),
                                  C.c_float(threshold), C.byref(n), C.byref(g))
    return float(r), int(n.value), float(g.value)


def set_weight_bits(bits):
    """0 = exact float interpolation weights (default); 8 = emulate the CUDA texture unit's 8-bit weights."""
    lib().orc_set_weight_bits(int(bits))


def ct_footprint(scn, beam):
    """Number of distinct CT voxels the tracer of this field reads (N_fp of the algorithmic-byte model)."""
    L = lib()
    L.orc_footprint_start.argtypes = [C.c_size_t]
    L.orc_footprint_stop.restype = C.c_longlong
    L.orc_footprint_start(int(scn.ct.size))
    scratch = np.zeros_like(scn.ct)
    run_field(scn, beam, scratch, keep_layers=False).close()
    return int(L.orc_footprint_stop())


def set_threads(n):
    lib().orc_set_threads(int(n))


def max_threads():
    return int(lib().orc_get_max_threads())
