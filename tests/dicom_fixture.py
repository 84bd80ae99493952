"""Writer of small synthetic DICOM inputs for tests of include/rtd_dicom.hpp: a CT series (one file per slice) and an RT Ion
Plan, in Explicit or Implicit VR Little Endian, sequences with defined or undefined length. Test infrastructure only —
nothing here is read by the product. (No DICOM file ships with the reference; ITK/GDCM/pydicom are absent, so these
fixtures are generated, not copied.)"""
import os
import struct

import numpy as np

EXPLICIT = "1.2.840.10008.1.2.1"
IMPLICIT = "1.2.840.10008.1.2"
_LONG = {"OB", "OW", "OF", "SQ", "UT", "UN"}


def _pad(b, vr):
    if len(b) % 2:
        b += b"\0" if vr in ("UI", "OB") else b" "
    return b


def _val(vr, v):
    if isinstance(v, bytes):
        return _pad(v, vr)
    if vr in ("DS", "IS"):
        vs = v if isinstance(v, (list, tuple, np.ndarray)) else [v]
        return _pad("\\".join(("%d" % x) if vr == "IS" else repr(float(x)) for x in vs).encode(), vr)
    if vr == "US":
        return struct.pack("<%dH" % len(np.atleast_1d(v)), *np.atleast_1d(v))
    if vr == "FL":
        return np.asarray(v, dtype="<f4").tobytes()
    return _pad(str(v).encode(), vr)


def elem(group, element, vr, value, explicit=True):
    """value: scalar / list / bytes, or a list of item-dicts (each a list of encoded elements) for SQ."""
    head = struct.pack("<HH", group, element)
    if vr == "SQ":
        items, undefined = value
        body = b""
        for it in items:
            payload = b"".join(it)
            if undefined:
                body += struct.pack("<HHI", 0xFFFE, 0xE000, 0xFFFFFFFF) + payload + struct.pack("<HHI", 0xFFFE, 0xE00D, 0)
            else:
                body += struct.pack("<HHI", 0xFFFE, 0xE000, len(payload)) + payload
        if undefined:
            body += struct.pack("<HHI", 0xFFFE, 0xE0DD, 0)
        length = 0xFFFFFFFF if undefined else len(body)
        if explicit:
            return head + b"SQ" + b"\0\0" + struct.pack("<I", length) + body
        return head + struct.pack("<I", length) + body
    data = _val(vr, value)
    if not explicit:
        return head + struct.pack("<I", len(data)) + data
    if vr in _LONG:
        return head + vr.encode() + b"\0\0" + struct.pack("<I", len(data)) + data
    return head + vr.encode() + struct.pack("<H", len(data)) + data


def _file(path, sop_class, dataset_bytes, syntax):
    meta = elem(0x0002, 0x0001, "OB", b"\0\1") + elem(0x0002, 0x0002, "UI", sop_class) + elem(0x0002, 0x0003, "UI", "1.2.3.4.5") \
        + elem(0x0002, 0x0010, "UI", syntax) + elem(0x0002, 0x0012, "UI", "1.2.3.99")
    meta = elem(0x0002, 0x0000, "UL", struct.pack("<I", len(meta))) + meta
    with open(path, "wb") as f:
        f.write(b"\0" * 128 + b"DICM" + meta + dataset_bytes)


def write_ct_series(directory, hu, spacing, origin, orientation=(1, 0, 0, 0, 1, 0), slope=1.0, intercept=-1024.0, syntax=EXPLICIT,
                    series_uid="1.2.826.0.1.3680043.8.498.1", shuffle=True, prefix="IM"):
    """hu: int array [nz][ny][nx] of Hounsfield units; spacing (dx, dy, dz); origin = position of voxel (0,0,0).
    Stored pixel = (HU - intercept) / slope as unsigned 16 bit. Files are written in shuffled order with arbitrary names."""
    ex = syntax == EXPLICIT
    nz, ny, nx = hu.shape
    rx, cx = np.array(orientation[:3], float), np.array(orientation[3:], float)
    nrm = np.cross(rx, cx)
    order = list(range(nz))
    if shuffle:
        order = order[1::2] + order[0::2][::-1]
    os.makedirs(directory, exist_ok=True)
    for n, k in enumerate(order):
        pos = np.array(origin, float) + nrm * spacing[2] * k
        stored = np.round((hu[k].astype(np.float64) - intercept) / slope).astype("<u2")
        ds = b"".join([
            elem(0x0008, 0x0016, "UI", "1.2.840.10008.5.1.4.1.1.2", ex), elem(0x0008, 0x0060, "CS", "CT", ex),
            elem(0x0018, 0x0050, "DS", spacing[2], ex),
            elem(0x0020, 0x000E, "UI", series_uid, ex), elem(0x0020, 0x0013, "IS", k + 1, ex),
            elem(0x0020, 0x0032, "DS", list(pos), ex), elem(0x0020, 0x0037, "DS", list(orientation), ex),
            elem(0x0028, 0x0002, "US", 1, ex), elem(0x0028, 0x0004, "CS", "MONOCHROME2", ex),
            elem(0x0028, 0x0010, "US", ny, ex), elem(0x0028, 0x0011, "US", nx, ex),
            elem(0x0028, 0x0030, "DS", [spacing[1], spacing[0]], ex),
            elem(0x0028, 0x0100, "US", 16, ex), elem(0x0028, 0x0101, "US", 16, ex), elem(0x0028, 0x0102, "US", 15, ex),
            elem(0x0028, 0x0103, "US", 0, ex),
            elem(0x0028, 0x1052, "DS", intercept, ex), elem(0x0028, 0x1053, "DS", slope, ex),
            elem(0x7FE0, 0x0010, "OW", stored.tobytes(), ex),
        ])
        _file(os.path.join(directory, "%s%04d.dcm" % (prefix, 7 * n + 3)), "1.2.840.10008.5.1.4.1.1.2", ds, syntax)
    with open(os.path.join(directory, "README.txt"), "w") as f:      # a non-DICOM file in the directory must be skipped
        f.write("not dicom\n")


def write_ion_plan(path, beams, syntax=EXPLICIT, undefined_length=True):
    """beams: list of dicts {name, gantry, couch, collimator, iso (3), vsad (2), layers: [ {energy, fwhm (2), x, y, w} ]};
    every layer is written as the usual pair of control points (weights, then the same positions with zero weights)."""
    ex = syntax == EXPLICIT
    beam_items = []
    for bi, b in enumerate(beams):
        cps = []
        idx = 0
        for li, lay in enumerate(b["layers"]):
            pos = np.stack([np.asarray(lay["x"], "f4"), np.asarray(lay["y"], "f4")], axis=1).ravel()
            for closing in (False, True):
                it = [elem(0x300A, 0x0112, "IS", idx, ex), elem(0x300A, 0x0114, "DS", lay["energy"], ex)]
                if li == 0 and not closing:
                    it += [elem(0x300A, 0x011E, "DS", b["gantry"], ex), elem(0x300A, 0x0120, "DS", b.get("collimator", 0.0), ex),
                           elem(0x300A, 0x0122, "DS", b.get("couch", 0.0), ex), elem(0x300A, 0x012C, "DS", list(b["iso"]), ex)]
                it += [elem(0x300A, 0x0392, "IS", len(lay["x"]), ex), elem(0x300A, 0x0394, "FL", pos, ex),
                       elem(0x300A, 0x0396, "FL", np.zeros(len(lay["x"]), "f4") if closing else np.asarray(lay["w"], "f4"), ex),
                       elem(0x300A, 0x0398, "FL", np.asarray(lay["fwhm"], "f4"), ex)]
                cps.append(it)
                idx += 1
        beam_items.append([
            elem(0x300A, 0x00C0, "IS", bi + 1, ex), elem(0x300A, 0x00C2, "LO", b["name"], ex), elem(0x300A, 0x00C6, "CS", "PROTON", ex),
            elem(0x300A, 0x030A, "FL", np.asarray(b["vsad"], "f4"), ex), elem(0x300A, 0x0110, "IS", len(cps), ex),
            elem(0x300A, 0x03A8, "SQ", (cps, undefined_length), ex),
        ])
    ds = b"".join([
        elem(0x0008, 0x0016, "UI", "1.2.840.10008.5.1.4.1.1.481.8", ex), elem(0x0008, 0x0060, "CS", "RTPLAN", ex),
        elem(0x300A, 0x0002, "SH", "synthetic", ex),
        elem(0x300A, 0x03A2, "SQ", (beam_items, undefined_length), ex),
    ])
    _file(path, "1.2.840.10008.5.1.4.1.1.481.8", ds, syntax)
