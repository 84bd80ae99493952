"""Synthetic inputs for the configurations named in BASELINE.json / SURVEY.md §8(d) (C1..C5).

Host-side mirror of what the reference's main() builds before calling cudaWrapperProtons
(src/main.cu:39-99,192-197): CT volume in HU+1000, geometry transforms, spot weights, energies, sigmas.
The reference fills weights with unseeded rand() (main.cu:80); a seeded generator is used here.
"""
import math

import numpy as np

from . import abi


class Float3AffineTransform:
    """y = m x + v (src/float3_affine_transform.cuh). Harness-side helper in float64, exported as float32."""

    def __init__(self, m=None, v=None):
        self.m = np.eye(3) if m is None else np.asarray(m, dtype=np.float64).reshape(3, 3)
        self.v = np.zeros(3) if v is None else np.asarray(v, dtype=np.float64).reshape(3)

    def inverse(self):
        mi = np.linalg.inv(self.m)
        return Float3AffineTransform(mi, -mi @ self.v)

    def transformPoint(self, p):
        return self.m @ np.asarray(p, dtype=np.float64) + self.v

    def as_abi(self):
        return abi.make_affine(self.m, self.v)


def concatFloat3AffineTransform(t1, t2):
    """Apply t1 then t2 (float3_affine_transform.cu:42-45)."""
    return Float3AffineTransform(t2.m @ t1.m, t2.m @ t1.v + t2.v)


class Float3IdxTransform:
    """y = x*delta + offset (src/float3_idx_transform.cuh)."""

    def __init__(self, delta=(1, 1, 1), offset=(0, 0, 0)):
        self.delta = np.asarray(delta, dtype=np.float64)
        self.offset = np.asarray(offset, dtype=np.float64)

    def as_abi(self):
        return abi.make_idx_transform(self.delta, self.offset)


class BeamSettings:
    """Host mirror of BeamSettings (src/beam_settings.h:31,101-109): the nine fields, same order."""

    def __init__(self, spotWeights, beamEnergies, spotSigmas, raySpacing, tracerSteps, sourceDist, spotIdxToGantry,
                 gantryToImIdx, gantryToDoseIdx):
        self.spotWeights = abi.f32(spotWeights)            # [L][ny][nx]
        assert self.spotWeights.ndim == 3
        self.beamEnergies = abi.f32(beamEnergies)
        self.spotSigmas = abi.f32(spotSigmas).reshape(-1, 2)
        assert self.beamEnergies.size == self.spotWeights.shape[0] == self.spotSigmas.shape[0]
        self.raySpacing = tuple(float(x) for x in raySpacing)
        self.tracerSteps = int(tracerSteps)
        self.sourceDist = tuple(float(x) for x in sourceDist)
        self.spotIdxToGantry = spotIdxToGantry
        self.gantryToImIdx = gantryToImIdx
        self.gantryToDoseIdx = gantryToDoseIdx

    def as_abi(self):
        b = abi.RtdBeam()
        b.spot_weights = abi.fptr(self.spotWeights)
        b.n_layers, b.spot_ny, b.spot_nx = (int(x) for x in self.spotWeights.shape)
        b.energies = abi.fptr(self.beamEnergies)
        b.spot_sigmas = abi.fptr(self.spotSigmas)
        b.ray_spacing[0], b.ray_spacing[1] = self.raySpacing
        b.tracer_steps = self.tracerSteps
        b.source_dist[0], b.source_dist[1] = self.sourceDist
        b.spot_idx_to_gantry = self.spotIdxToGantry.as_abi()
        b.gantry_to_im_idx = self.gantryToImIdx.as_abi()
        b.gantry_to_dose_idx = self.gantryToDoseIdx.as_abi()
        return b


def beams_abi(beams):
    arr = (abi.RtdBeam * len(beams))()
    for i, b in enumerate(beams):
        arr[i] = b.as_abi()
    return arr


class Scenario:
    def __init__(self, name, luts, ct, spacing, beams, description=""):
        self.name = name
        self.luts = luts
        self.ct = abi.f32(ct)                       # [Z][Y][X] HU+1000
        self.dims = (self.ct.shape[2], self.ct.shape[1], self.ct.shape[0])   # (x, y, z) like uint3
        self.spacing = tuple(float(s) for s in spacing)
        self.beams = beams
        self.description = description

    @property
    def n_voxels(self):
        return int(self.ct.size)


def rotation_y(deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    # snap the exact quarter turns so G090/G180/G270 are axis-aligned
    c, s = (round(c) if abs(c - round(c)) < 1e-12 else c), (round(s) if abs(s - round(s)) < 1e-12 else s)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def water_cube_energies(luts, n_layers, e0=118.12, e1=172.51):
    """Energies and spot sigmas of the reference water cube (main.cu:86-99)."""
    step = (e1 - e0) / float(n_layers - 1) if n_layers > 1 else 0.0
    energies = np.array([e0 + i * step for i in range(n_layers)], dtype=np.float32)
    # findDecimalOrdered + vectorInterpolate on peakDepths (harness copy in float64; inputs only)
    idx = np.interp(energies, luts.energiesPerU, np.arange(luts.nEnergies))
    peak = np.interp(idx, np.arange(luts.nEnergies), luts.peakDepths)
    sig = 2.3 + 290.0 / (peak + 15.0)
    return energies, np.stack([sig, sig], axis=1).astype(np.float32)


def _geometry(n, voxel, origin, gantry_deg=0.0, gantry_rot=None):
    imIdxToWorld = Float3AffineTransform(np.eye(3) * voxel, origin)
    worldToImIdx = imIdxToWorld.inverse()
    gantryToWorld = Float3AffineTransform(rotation_y(gantry_deg) if gantry_rot is None else np.asarray(gantry_rot, dtype=float),
                                          (0.0, 0.0, 0.0))
    return concatFloat3AffineTransform(gantryToWorld, worldToImIdx)     # main.cu:57


def make_field(luts, n, voxel, origin, gantry_deg, spots, pitch, n_layers, seed, source_dist=(math.inf, math.inf),
               steps=512, ray_spacing=(1.0, 1.0), start_z=128.0, step_len=1.0, weight_lo=90.0, weight_span=10.0, gantry_rot=None):
    """gantry_rot: a 3x3 gantry -> world rotation replacing the rotation about the world Y axis by gantry_deg."""
    gantryToImIdx = _geometry(n, voxel, origin, gantry_deg, gantry_rot)
    nx, ny = (spots, spots) if np.isscalar(spots) else (int(spots[0]), int(spots[1]))     # a pair: (columns, rows) of the spot map
    px, py = (pitch, pitch) if np.isscalar(pitch) else (float(pitch[0]), float(pitch[1]))
    spotIdxToGantry = Float3IdxTransform((px, py, -step_len), (-0.5 * (nx - 1) * px, -0.5 * (ny - 1) * py, start_z))
    rng = np.random.default_rng(seed)
    weights = (weight_lo + weight_span * rng.random((n_layers, ny, nx))).astype(np.float32)
    energies, sigmas = water_cube_energies(luts, n_layers)
    return BeamSettings(weights, energies, sigmas, ray_spacing, steps, source_dist, spotIdxToGantry, gantryToImIdx,
                        gantryToImIdx)


def water_cube(luts, n=256, n_layers=20, spots=33, pitch=3.0, seed=1234, gantry_deg=0.0, source_dist=(math.inf, math.inf),
               steps=512, ray_spacing=(1.0, 1.0)):
    """C2 = the reference's WATER_CUBE_TEST (main.cu:39-43,61-62,74-99) for n=256; C1 for n=128, n_layers=1."""
    voxel = 256.0 / n
    ct = np.full((n, n, n), 1000.0, dtype=np.float32)
    origin = (-128.0, -128.0, -256.0 + 150.0)
    beam = make_field(luts, n, voxel, origin, gantry_deg, spots, pitch, n_layers, seed, source_dist, steps, ray_spacing)
    return Scenario("water%d_L%d" % (n, n_layers), luts, ct, (voxel,) * 3, [beam],
                    "water cube %d^3, %d layer(s), %dx%d spots" % (n, n_layers, spots, spots))


def hetero_phantom(n, seed=7, noise=20.0):
    """Analytic heterogeneous phantom (HU+1000 in [0,3071]) on the 256 mm cube with origin (-128,-128,-106):
    an elliptical body (water 1000) in air (0), a bone sphere 2200, a lung slab 300, an air cavity, seeded noise."""
    voxel = 256.0 / n
    ax = (np.arange(n, dtype=np.float32) * voxel).astype(np.float32)
    x = (ax - 128.0)[None, None, :]
    y = (ax - 128.0)[None, :, None]
    z = (ax - 106.0)[:, None, None]
    ct = np.zeros((n, n, n), dtype=np.float32)
    body = (x / 115.0) ** 2 + ((z - 22.0) / 105.0) ** 2 <= 1.0
    body = np.broadcast_to(body, ct.shape)
    ct[body] = 1000.0
    lung = (np.abs(x + 45.0) < 22.0) & (np.abs(z - 50.0) < 30.0) & (np.abs(y) < 60.0)
    ct[np.broadcast_to(lung, ct.shape) & body] = 300.0
    bone = (x - 30.0) ** 2 + y ** 2 + (z - 40.0) ** 2 < 24.0 ** 2
    ct[np.broadcast_to(bone, ct.shape)] = 2200.0
    cav = (x + 10.0) ** 2 + (y - 15.0) ** 2 + (z - 75.0) ** 2 < 9.0 ** 2
    ct[np.broadcast_to(cav, ct.shape)] = 5.0
    rng = np.random.default_rng(seed)
    # noise only inside the body, generated slice-wise to bound memory
    for k in range(n):
        sl = ct[k]
        m = sl > 100.0
        sl[m] += (rng.random(int(m.sum()), dtype=np.float32) * 2.0 - 1.0) * noise
    np.clip(ct, 0.0, 3071.0, out=ct)
    return ct, voxel


def hetero_ct(luts, n=512, n_fields=1, spots=10, pitch=6.0, n_layers=20, seed=99, source_dist=(math.inf, math.inf),
              angles=None, steps=512, ct=None, gantry_rot=None, ray_spacing=(1.0, 1.0)):
    """C3 (n=512, 1 field), C4 (n=512, 4 fields at 0/90/180/270), C5 (n=768, 8 fields every 45 deg)."""
    if ct is None:
        ct, voxel = hetero_phantom(n)
    else:
        voxel = 256.0 / n
    origin = (-128.0, -128.0, -106.0)
    if angles is None:
        angles = [i * 360.0 / n_fields for i in range(n_fields)]
    beams = [make_field(luts, n, voxel, origin, a, spots, pitch, n_layers, seed + 17 * i, source_dist, steps, ray_spacing=ray_spacing, gantry_rot=gantry_rot)
             for i, a in enumerate(angles)]
    snx, sny = (spots, spots) if np.isscalar(spots) else spots
    return Scenario("hetero%d_F%d" % (n, len(beams)), luts, ct, (voxel,) * 3, beams,
                    "heterogeneous CT %d^3, %d field(s), %dx%dx%d spots" % (n, len(beams), snx, sny, n_layers))
