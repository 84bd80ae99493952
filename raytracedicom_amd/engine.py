"""Host-side binding of the HIP dose engine (raytracedicom_amd/librtd_hip.so) through its C ABI (include/rtd.h).

This is the product path: every call goes to the hand-written HIP kernels. There is NO CPU fallback — if the
shared library is missing or no GPU is present the calls raise RtdError loudly.

`cudaWrapperProtons` mirrors the reference entry point of the same name (src/kernel_wrapper.cuh:161): same
argument meaning, dose accumulated in place, log text written to the stream argument, errors raised.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtd_hip.so")
_LIB = None

_FETCH_DTYPES = {"first_inside": np.int32, "first_outside": np.int32, "first_passive": np.int32,
                 "eff_radius": np.int32, "tile_radius": np.uint8, "fill_debug": np.int64, "sweep_debug": np.int64, "uniform_debug": np.int64, "sweep_big_debug": np.int64, "scan_debug": np.int64}


class RtdError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rtd status %d: %s" % (status, message))
        self.status = status


def build(verbose=False):
    """Compile the HIP engine for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "all"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def lib():
    """Load librtd_hip.so. Raises if it has not been built: the product never substitutes another path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RtdError(abi.RTD_ERR_NO_DEVICE, "%s not built (run __graft_entry__.build() or make -C raytracedicom_amd/csrc)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
        u3 = C.POINTER(C.c_uint32)
        L.rtd_abi_version.restype = C.c_uint32
        L.rtd_last_error.restype = C.c_char_p
        L.rtd_last_error.argtypes = [vp]
        L.rtd_global_error.restype = C.c_char_p
        L.rtd_create.argtypes = [C.c_int, vpp]
        L.rtd_destroy.argtypes = [vp]
        L.rtd_set_options.argtypes = [vp, C.POINTER(abi.RtdOptions)]
        L.rtd_set_luts.argtypes = [vp, C.POINTER(abi.RtdLuts)]
        L.rtd_load_luts_dir.argtypes = [vp, C.c_char_p, C.c_int]
        L.rtd_set_ct.argtypes = [vp, abi.c_float_p, u3]
        L.rtd_set_ct_device.argtypes = [vp, vp, u3]
        L.rtd_compute.argtypes = [vp, C.POINTER(abi.RtdBeam), C.c_int, abi.c_float_p, u3, C.POINTER(abi.RtdTiming)]
        L.rtd_field_create.argtypes = [vp, C.POINTER(abi.RtdBeam), u3, vpp]
        L.rtd_field_compute.argtypes = [vp, vp, vp]
        L.rtd_field_finish.argtypes = [vp, vp, C.POINTER(abi.RtdTiming), C.POINTER(abi.RtdFieldInfo)]
        L.rtd_field_clear_dose.argtypes = [vp, vp, vp]
        L.rtd_field_destroy.argtypes = [vp, vp]
        L.rtd_field_fetch.argtypes = [vp, vp, C.c_char_p, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        i3 = C.POINTER(C.c_int32)
        L.rtd_field_compute_bev.argtypes = [vp, vp]
        L.rtd_field_transfer.argtypes = [vp, vp, vp, i3, i3]
        L.rtd_field_transfer_init.argtypes = [vp, vp, vp, i3, i3]
        L.rtd_fields_transfer_init.argtypes = [vp, C.POINTER(C.c_void_p), C.c_uint32, vp, i3, i3]
        L.rtd_field_wait_plan.argtypes = [vp, vp, C.POINTER(abi.RtdFieldInfo), C.POINTER(C.c_size_t)]
        L.rtd_bev_message_bound.argtypes = [vp, vp]
        L.rtd_bev_message_bound.restype = C.c_size_t
        L.rtd_field_export_bev.argtypes = [vp, vp, vp, C.c_size_t]
        L.rtd_field_create_remote.argtypes = [vp, C.POINTER(abi.RtdBeam), u3, vpp]
        L.rtd_field_attach_bev.argtypes = [vp, vp, vp]
        L.rtd_field_clear_dose_box.argtypes = [vp, vp, vp, i3, i3]
        L.rtd_field_release.argtypes = [vp, vp]
        L.rtd_host_register.argtypes = [vp, C.c_size_t]
        L.rtd_host_unregister.argtypes = [vp]
        L.rtd_plan_create.argtypes = [C.POINTER(C.c_int), C.c_int, vpp]
        L.rtd_plan_destroy.argtypes = [vp]
        L.rtd_plan_last_error.argtypes = [vp]
        L.rtd_plan_last_error.restype = C.c_char_p
        L.rtd_plan_set_options.argtypes = [vp, C.POINTER(abi.RtdOptions)]
        L.rtd_plan_set_luts.argtypes = [vp, C.POINTER(abi.RtdLuts)]
        L.rtd_plan_load_luts_dir.argtypes = [vp, C.c_char_p, C.c_int]
        L.rtd_plan_set_ct.argtypes = [vp, abi.c_float_p, u3]
        L.rtd_plan_set_ct_deferred.argtypes = [vp, abi.c_float_p, u3]
        L.rtd_set_ct_deferred.argtypes = [vp, abi.c_float_p, u3]
        L.rtd_plan_compute.argtypes = [vp, C.POINTER(abi.RtdBeam), C.c_int, abi.c_float_p, u3, C.POINTER(abi.RtdTiming),
                                       C.POINTER(abi.RtdPlanTiming)]
        L.rtd_device_alloc.argtypes = [vp, C.c_size_t, vpp]
        L.rtd_device_free.argtypes = [vp, vp]
        L.rtd_device_zero.argtypes = [vp, vp, C.c_size_t]
        L.rtd_copy_to_device.argtypes = [vp, vp, vp, C.c_size_t]
        L.rtd_copy_to_host.argtypes = [vp, vp, vp, C.c_size_t]
        L.rtd_sync.argtypes = [vp]
        L.rtd_stream.argtypes = [vp]
        L.rtd_stream.restype = vp
        L.rtd_set_stream.argtypes = [vp, vp]
        _LIB = L
    return _LIB


class Field:
    """One beam prepared on the device (rtd_field_*)."""

    def __init__(self, eng, beam, dose_dims, remote=False):
        self.eng = eng
        self._beam = beam            # keeps the numpy arrays alive
        self._h = C.c_void_p()
        ba = beam.as_abi()
        create = lib().rtd_field_create_remote if remote else lib().rtd_field_create
        eng._check(create(eng._h, C.byref(ba), abi.uint3(dose_dims), C.byref(self._h)))
        self.remote = remote
        self.computed = False        # a compute() has been launched (clear_dose / finish are valid)

    @staticmethod
    def _clip(lo, hi):
        if lo is None:
            return None, None
        return (C.c_int32 * 3)(*[int(v) for v in lo]), (C.c_int32 * 3)(*[int(v) for v in hi])

    def compute_bev(self):
        """All kernels up to the beam's-eye-view dose; asynchronous."""
        self.eng._check(lib().rtd_field_compute_bev(self.eng._h, self._h))
        self.computed = True

    def transfer(self, dev_dose, clip_min=None, clip_max=None):
        """Fan -> dose-grid transfer of the field's BEV dose (its own or an attached slab) into dev_dose, optionally restricted
        to the inclusive dose-index box [clip_min, clip_max]; asynchronous."""
        lo, hi = self._clip(clip_min, clip_max)
        self.eng._check(lib().rtd_field_transfer(self.eng._h, self._h, C.c_void_p(int(dev_dose)), lo, hi))

    def transfer_init(self, dev_dose, clip_min=None, clip_max=None):
        """transfer() for the first field of a plan: the field's dose box is written (dose or zero), not accumulated into."""
        lo, hi = self._clip(clip_min, clip_max)
        self.eng._check(lib().rtd_field_transfer_init(self.eng._h, self._h, C.c_void_p(int(dev_dose)), lo, hi))

    def wait_plan(self):
        """(info, packed_bytes) once the device-side plan of the field is known (the superposition may still be running)."""
        i, n = abi.RtdFieldInfo(), C.c_size_t(0)
        self.eng._check(lib().rtd_field_wait_plan(self.eng._h, self._h, C.byref(i), C.byref(n)))
        return i.as_dict(), int(n.value)

    def message_bound(self):
        return int(lib().rtd_bev_message_bound(self.eng._h, self._h))

    def export_bev(self, dev_buf, capacity):
        """Pack [state record | non-zero block of the BEV dose] into dev_buf (device pointer); asynchronous."""
        self.eng._check(lib().rtd_field_export_bev(self.eng._h, self._h, C.c_void_p(int(dev_buf)), int(capacity)))

    def attach_bev(self, dev_buf):
        """Remote field: sample the message at dev_buf (device pointer; not copied)."""
        self.eng._check(lib().rtd_field_attach_bev(self.eng._h, self._h, C.c_void_p(int(dev_buf))))
        self.computed = True

    def clear_dose_box(self, dev_dose, clip_min=None, clip_max=None):
        lo, hi = self._clip(clip_min, clip_max)
        self.eng._check(lib().rtd_field_clear_dose_box(self.eng._h, self._h, C.c_void_p(int(dev_dose)), lo, hi))

    def compute(self, dev_dose):
        """Launch all kernels of the field; asynchronous. dev_dose: device pointer (int) of the dose volume."""
        self.eng._check(lib().rtd_field_compute(self.eng._h, self._h, C.c_void_p(int(dev_dose))))
        self.computed = True

    def clear_dose(self, dev_dose):
        """Zero the voxels of dev_dose that the last compute() of this field could have changed; asynchronous."""
        self.eng._check(lib().rtd_field_clear_dose(self.eng._h, self._h, C.c_void_p(int(dev_dose))))

    def finish(self):
        t, i = abi.RtdTiming(), abi.RtdFieldInfo()
        self.eng._check(lib().rtd_field_finish(self.eng._h, self._h, C.byref(t), C.byref(i)))
        return t.as_dict(), i.as_dict()

    def fetch(self, name):
        n = C.c_size_t(0)
        self.eng._check(lib().rtd_field_fetch(self.eng._h, self._h, name.encode(), None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint8)
        self.eng._check(lib().rtd_field_fetch(self.eng._h, self._h, name.encode(), out.ctypes.data_as(C.c_void_p), n.value, None))
        return out.view(_FETCH_DTYPES.get(name, np.float32))

    def destroy(self):
        if self._h:
            lib().rtd_field_destroy(self.eng._h, self._h)
            self._h = C.c_void_p()

    def release(self):
        """Like destroy(), but the device workspace stays with the engine for the next field of the same shape."""
        if self._h:
            lib().rtd_field_release(self.eng._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Engine:
    """One rtd_handle: one GPU, one stream, resident CT and LUTs."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        st = lib().rtd_create(int(device_id), C.byref(self._h))
        if st != 0:
            raise RtdError(st, lib().rtd_global_error().decode())
        self._keep = []

    def _check(self, st):
        if st != 0:
            raise RtdError(st, lib().rtd_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().rtd_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_options(self, opt):
        self._check(lib().rtd_set_options(self._h, C.byref(opt)))

    def set_luts(self, es):
        la = es.as_abi()
        self._check(lib().rtd_set_luts(self._h, C.byref(la)))

    def load_luts_dir(self, directory, water_cube_test=False):
        self._check(lib().rtd_load_luts_dir(self._h, directory.encode(), int(water_cube_test)))

    def set_ct(self, ct, deferred=False):
        """deferred: rtd_set_ct_deferred — ct (C-contiguous float32, kept alive here) must stay unchanged until the computes that use
        it are done; each field then uploads only the box of it that its rays cross."""
        if deferred:
            assert ct.dtype == np.float32 and ct.flags["C_CONTIGUOUS"]
            self._ct_keep = ct
            self._check(lib().rtd_set_ct_deferred(self._h, abi.fptr(ct), abi.uint3((ct.shape[2], ct.shape[1], ct.shape[0]))))
            return
        ct = abi.f32(ct)
        self._check(lib().rtd_set_ct(self._h, abi.fptr(ct), abi.uint3((ct.shape[2], ct.shape[1], ct.shape[0]))))

    def set_ct_device(self, dev_ptr, dims):
        self._check(lib().rtd_set_ct_device(self._h, C.c_void_p(int(dev_ptr)), abi.uint3(dims)))

    def compute(self, beams, dose):
        """rtd_compute: reference-shaped, accumulates into the host array dose ([Z][Y][X] float32)."""
        from .scenarios import beams_abi
        assert dose.dtype == np.float32 and dose.flags["C_CONTIGUOUS"]
        ba = beams_abi(beams)
        tm = (abi.RtdTiming * max(1, len(beams)))()
        self._check(lib().rtd_compute(self._h, ba, len(beams), abi.fptr(dose), abi.uint3((dose.shape[2], dose.shape[1], dose.shape[0])), tm))
        return [tm[i].as_dict() for i in range(len(beams))]

    def create_field(self, beam, dose_dims, remote=False):
        return Field(self, beam, dose_dims, remote=remote)

    def transfer_fields_init(self, fields, dev_dose, box_min=None, box_max=None):
        """rtd_fields_transfer_init: every voxel of the inclusive dose-index box is written with 0 + fields[0] + fields[1] + ...
        in one launch (bit for bit the loop of Field.transfer over the fields into a zeroed box); asynchronous."""
        arr = (C.c_void_p * len(fields))(*[f._h for f in fields])
        lo, hi = Field._clip(box_min, box_max)
        self._check(lib().rtd_fields_transfer_init(self._h, arr, len(fields), C.c_void_p(int(dev_dose)), lo, hi))

    # device buffers owned by the handle
    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._check(lib().rtd_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, p):
        self._check(lib().rtd_device_free(self._h, C.c_void_p(p)))

    def device_zero(self, p, nbytes):
        self._check(lib().rtd_device_zero(self._h, C.c_void_p(p), nbytes))

    def to_device(self, p, arr):
        arr = np.ascontiguousarray(arr)
        self._check(lib().rtd_copy_to_device(self._h, C.c_void_p(p), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def to_host(self, arr, p):
        assert arr.flags["C_CONTIGUOUS"]
        self._check(lib().rtd_copy_to_host(self._h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(p), arr.nbytes))

    def sync(self):
        self._check(lib().rtd_sync(self._h))

    def stream(self):
        return lib().rtd_stream(self._h)

    def set_stream(self, s):
        self._check(lib().rtd_set_stream(self._h, C.c_void_p(s) if s else None))


class Plan:
    """rtd_plan_*: the reference-shaped call on several GPUs of one process (one host thread per device)."""

    def __init__(self, device_ids):
        self._h = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        st = lib().rtd_plan_create(ids, len(device_ids), C.byref(self._h))
        if st != 0:
            raise RtdError(st, lib().rtd_global_error().decode())

    def _check(self, st):
        if st != 0:
            raise RtdError(st, lib().rtd_plan_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().rtd_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_options(self, opt):
        self._check(lib().rtd_plan_set_options(self._h, C.byref(opt)))

    def set_luts(self, es):
        la = es.as_abi()
        self._check(lib().rtd_plan_set_luts(self._h, C.byref(la)))

    def set_ct(self, ct, deferred=False):
        """deferred: rtd_plan_set_ct_deferred — ct (a C-contiguous float32 array, kept alive here) must stay unchanged until the
        computes that use it are done; each beam then uploads only the box of it that its rays cross."""
        if deferred:
            assert ct.dtype == np.float32 and ct.flags["C_CONTIGUOUS"]
            self._ct_keep = ct
            self._check(lib().rtd_plan_set_ct_deferred(self._h, abi.fptr(ct), abi.uint3((ct.shape[2], ct.shape[1], ct.shape[0]))))
            return
        ct = abi.f32(ct)
        self._check(lib().rtd_plan_set_ct(self._h, abi.fptr(ct), abi.uint3((ct.shape[2], ct.shape[1], ct.shape[0]))))

    def compute(self, beams, dose):
        """All beams accumulated into the host array dose ([Z][Y][X] float32); returns (per-beam timings, plan timing)."""
        from .scenarios import beams_abi
        assert dose.dtype == np.float32 and dose.flags["C_CONTIGUOUS"]
        ba = beams_abi(beams)
        tm = (abi.RtdTiming * max(1, len(beams)))()
        pt = abi.RtdPlanTiming()
        self._check(lib().rtd_plan_compute(self._h, ba, len(beams), abi.fptr(dose), abi.uint3((dose.shape[2], dose.shape[1], dose.shape[0])), tm,
                                           C.byref(pt)))
        return [tm[i].as_dict() for i in range(len(beams))], pt.as_dict()


def host_register(arr):
    """Page-lock a numpy array (HostPinnedImage3D's cudaHostRegister, host_image_3d.cuh:23-32)."""
    st = lib().rtd_host_register(arr.ctypes.data_as(C.c_void_p), arr.nbytes)
    if st != 0:
        raise RtdError(st, lib().rtd_global_error().decode())


def host_unregister(arr):
    lib().rtd_host_unregister(arr.ctypes.data_as(C.c_void_p))


def cudaWrapperProtons(imVol, doseVol, beams, iddData, outStream=None, device_id=0, options=None):
    """Drop-in for the reference's cudaWrapperProtons (src/kernel_wrapper.cuh:161).

    imVol   [Z][Y][X] float32 HU+1000; doseVol [Z][Y][X] float32, accumulated in place;
    beams   list of scenarios.BeamSettings; iddData luts.EnergyStruct; outStream file-like for the log text.
    """
    out = outStream if outStream is not None else sys.stdout
    opt = options or abi.default_options()
    with Engine(device_id) as eng:
        eng.set_options(opt)
        eng.set_luts(iddData)
        eng.set_ct(imVol)
        timings = eng.compute(beams, doseVol)
    total = sum(t["total_ms"] for t in timings)
    if opt.fine_grained_timing:
        for i, t in enumerate(timings):   # bucket names of kernel_wrapper.cu:1298-1307
            out.write("    Calculating field no. %d\n" % i)
            out.write("        Time to trace rays: %g ms\n" % t["raytracing_ms"])
            out.write("        Time preparing data for loop over energies: %g ms\n" % t["prepare_energy_loop_ms"])
            out.write("        Time depositing IDD and calculating sigma: %g ms\n" % t["fill_idd_sigma_ms"])
            out.write("        Time preparing for superposition: %g ms\n" % t["prepare_superp_ms"])
            out.write("        Time executing superposition: %g ms\n" % t["superp_ms"])
            out.write("        Kernel time to transform voxels: %g ms\n\n" % t["transforming_ms"])
    out.write("    Total global execution time (excluding GPU initialisation): %g ms.\n\n" % total)
    return timings
