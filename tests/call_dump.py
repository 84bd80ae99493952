"""Reader of the RTD_DUMP_CALL file written by the C++ shim (include/rtd_wrapper.hpp): the marshalled call — CT, dims and every
rtd_beam with its arrays — exactly as it crossed the C ABI. Lets a test run the oracle on what the CLI really computed."""
import ctypes as C

import numpy as np

from raytracedicom_amd import abi, scenarios


class _Raw:
    def __init__(self, struct):
        self._s = struct

    def as_abi(self):
        return self._s


def read(path, luts, spacing=(1.0, 1.0, 1.0)):
    """-> scenarios.Scenario holding the dumped CT and beams (dose grid = CT grid in every CLI mode)."""
    buf = open(path, "rb").read()
    head = np.frombuffer(buf, dtype=np.uint32, count=8)
    assert head[0] == 0x52544443, "not an RTD_DUMP_CALL file"
    im, dd, nb = [int(v) for v in head[1:4]], [int(v) for v in head[4:7]], int(head[7])
    assert im == dd
    off = 32
    n = im[0] * im[1] * im[2]
    ct = np.frombuffer(buf, dtype=np.float32, count=n, offset=off).reshape(im[2], im[1], im[0]).copy()
    off += 4 * n
    beams = []
    for _ in range(nb):
        nx, ny, L, steps = [int(v) for v in np.frombuffer(buf, dtype=np.uint32, count=4, offset=off)]
        off += 16
        ray = np.frombuffer(buf, dtype=np.float32, count=2, offset=off); off += 8
        sad = np.frombuffer(buf, dtype=np.float32, count=2, offset=off); off += 8
        sitg = abi.RtdIdxTransform.from_buffer_copy(buf, off); off += C.sizeof(abi.RtdIdxTransform)
        gtii = abi.RtdAffine.from_buffer_copy(buf, off); off += C.sizeof(abi.RtdAffine)
        gtdi = abi.RtdAffine.from_buffer_copy(buf, off); off += C.sizeof(abi.RtdAffine)
        energies = np.frombuffer(buf, dtype=np.float32, count=L, offset=off).copy(); off += 4 * L
        sigmas = np.frombuffer(buf, dtype=np.float32, count=2 * L, offset=off).copy(); off += 8 * L
        w = np.frombuffer(buf, dtype=np.float32, count=L * ny * nx, offset=off).reshape(L, ny, nx).copy(); off += 4 * L * ny * nx
        beams.append(scenarios.BeamSettings(w, energies, sigmas, (float(ray[0]), float(ray[1])), steps, (float(sad[0]), float(sad[1])),
                                            _Raw(sitg), _Raw(gtii), _Raw(gtdi)))
    assert off == len(buf)
    return scenarios.Scenario("dumped_call", luts, ct, spacing, beams)
