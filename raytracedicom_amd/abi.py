"""ctypes image of include/rtd.h (the C ABI of the dose engine).

Each Structure mirrors one POD of include/rtd.h field for field; see that header for the reference
type each one replaces (BeamSettings src/beam_settings.h:101-109, EnergyStruct src/energy_struct.h:13-31,
Float3AffineTransform, Float3IdxTransform).
"""
import ctypes as C

RTD_ABI_VERSION = 3     # include/rtd.h

import numpy as np

RTD_OK = 0
RTD_ERR_INVALID_ARG = -1
RTD_ERR_HIP = -2
RTD_ERR_RADIUS_OVERFLOW = -3
RTD_ERR_NOT_READY = -4
RTD_ERR_IO = -5
RTD_ERR_NO_DEVICE = -6
RTD_NUC_OFF, RTD_NUC_SOUKUP, RTD_NUC_FLUKA, RTD_NUC_GAUSS_FIT = 0, 1, 2, 3

c_float_p = C.POINTER(C.c_float)


class RtdAffine(C.Structure):
    _fields_ = [("m", C.c_float * 9), ("v", C.c_float * 3)]


class RtdIdxTransform(C.Structure):
    _fields_ = [("delta", C.c_float * 3), ("offset", C.c_float * 3)]


class RtdBeam(C.Structure):
    _fields_ = [
        ("spot_weights", c_float_p),
        ("spot_nx", C.c_uint32), ("spot_ny", C.c_uint32), ("n_layers", C.c_uint32),
        ("energies", c_float_p),
        ("spot_sigmas", c_float_p),
        ("ray_spacing", C.c_float * 2),
        ("tracer_steps", C.c_uint32),
        ("source_dist", C.c_float * 2),
        ("spot_idx_to_gantry", RtdIdxTransform),
        ("gantry_to_im_idx", RtdAffine),
        ("gantry_to_dose_idx", RtdAffine),
    ]


class RtdLuts(C.Structure):
    _fields_ = [
        ("n_energy_samples", C.c_int32), ("n_energies", C.c_int32),
        ("energies_per_u", c_float_p), ("peak_depths", c_float_p), ("scale_facts", c_float_p),
        ("cidd_matrix", c_float_p),
        ("n_density_samples", C.c_int32), ("density_scale_fact", C.c_float), ("density_vector", c_float_p),
        ("n_sp_samples", C.c_int32), ("sp_scale_fact", C.c_float), ("sp_vector", c_float_p),
        ("n_rrl_samples", C.c_int32), ("rrl_scale_fact", C.c_float), ("rrl_vector", c_float_p),
        ("nuc_weight_matrix", c_float_p), ("nuc_sq_sigma_matrix", c_float_p),
    ]


class RtdOptions(C.Structure):
    _fields_ = [
        ("dose_to_water", C.c_int32), ("nozzle", C.c_int32),
        ("bp_depth_cutoff", C.c_float), ("conv_sigma_cutoff", C.c_float),
        ("ks_sigma_cutoff", C.c_float), ("ray_weight_cutoff", C.c_float),
        ("fine_grained_timing", C.c_int32), ("nuclear_corr", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


class RtdTiming(C.Structure):
    _fields_ = [
        ("raytracing_ms", C.c_float), ("prepare_energy_loop_ms", C.c_float), ("fill_idd_sigma_ms", C.c_float),
        ("prepare_superp_ms", C.c_float), ("superp_ms", C.c_float), ("transforming_ms", C.c_float),
        ("total_ms", C.c_float), ("superp_launches", C.c_int32), ("superp_kernel_ms", C.c_float),
        ("ray_dims", C.c_uint32 * 2), ("steps", C.c_uint32), ("n_layers", C.c_uint32), ("transfer_voxels", C.c_int64),
        ("reserved", C.c_int32 * 2),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}
        d["ray_dims"] = list(d["ray_dims"])
        return d


class RtdFieldInfo(C.Structure):
    _fields_ = [
        ("ray_dims", C.c_uint32 * 3), ("ray_offset", C.c_float * 3), ("ray_res", C.c_float * 3),
        ("beam_first_inside", C.c_int32), ("beam_first_outside", C.c_int32),
        ("beam_first_guaranteed_passive", C.c_int32), ("beam_first_calculated_passive", C.c_int32),
        ("bbox_min", C.c_int32 * 3), ("bbox_max", C.c_int32 * 3),
        ("live_steps", C.c_int64), ("max_radius", C.c_int32),
        ("dose_box_min", C.c_int32 * 3), ("dose_box_max", C.c_int32 * 3), ("uniform_sigma", C.c_int32),
    ]

    def as_dict(self):
        out = {}
        for k, _ in self._fields_:
            if k == "reserved":
                continue
            v = getattr(self, k)
            out[k] = list(v) if hasattr(v, "__len__") else v
        return out


class RtdPlanTiming(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("upload_ms", C.c_float), ("bev_ms", C.c_float), ("exchange_ms", C.c_float),
                ("transfer_ms", C.c_float), ("download_ms", C.c_float), ("n_devices", C.c_int32), ("reserved", C.c_int32 * 3)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


def default_options():
    """Reference defaults of the compile-time switches (CMakeLists.txt:36-79)."""
    o = RtdOptions()
    o.dose_to_water = 1
    o.nozzle = 1
    o.bp_depth_cutoff = 1.05
    o.conv_sigma_cutoff = 3.0
    o.ks_sigma_cutoff = 3.0
    o.ray_weight_cutoff = 1.0
    o.fine_grained_timing = 0
    o.nuclear_corr = 0
    return o


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


def make_affine(m, v):
    """Float3AffineTransform: m 3x3 row-major, v offset."""
    a = RtdAffine()
    mm = np.asarray(m, dtype=np.float32).reshape(9)
    vv = np.asarray(v, dtype=np.float32).reshape(3)
    for i in range(9):
        a.m[i] = float(mm[i])
    for i in range(3):
        a.v[i] = float(vv[i])
    return a


def make_idx_transform(delta, offset):
    t = RtdIdxTransform()
    for i in range(3):
        t.delta[i] = float(np.float32(delta[i]))
        t.offset[i] = float(np.float32(offset[i]))
    return t


def uint3(dims):
    return (C.c_uint32 * 3)(int(dims[0]), int(dims[1]), int(dims[2]))
