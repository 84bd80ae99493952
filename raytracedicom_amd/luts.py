"""Look-up tables of the beam model in the reference's EnergyStruct layout (src/energy_struct.h:13-31).

* read_lut_dir / write_lut_dir: the reference's whitespace-separated text layout
  (src/energy_reader.cpp:12-101). The engine itself loads that layout in C++
  through rtd_load_luts_dir; this Python reader exists for the harness (tests, bench) and for writing
  synthetic tables in the same layout.
* synth_luts: physically plausible synthetic tables (analytic Bragg curves, Schneider-like HU->density
  and HU->SP ramps, 1/X0(density)). The reference's own LUT files are GPL data and do not ship with this
  repository or travel to the GPU box; benchmarks and GPU tests run on these synthetic tables.
"""
import os

import numpy as np

from . import abi

FILES = {
    "cidd": "proton_cumul_ddd_data.txt",
    "density": "density_Schneider2000_adj.txt",
    "sp": "HU_to_SP_H&N_adj.txt",
    "rrl": "radiation_length.txt",
    "rrl_water": "radiation_length_inc_water.txt",
}


class EnergyStruct:
    """Host mirror of the reference's EnergyStruct (energy_struct.h:13-31), arrays as float32 numpy."""

    def __init__(self, energiesPerU, peakDepths, scaleFacts, ciddMatrix, densityScaleFact, densityVector,
                 spScaleFact, spVector, rRlScaleFact, rRlVector, nucWeightMatrix=None, nucSqSigmaMatrix=None):
        self.energiesPerU = abi.f32(energiesPerU)
        self.peakDepths = abi.f32(peakDepths)
        self.scaleFacts = abi.f32(scaleFacts)
        self.nEnergies = int(self.energiesPerU.size)
        self.ciddMatrix = abi.f32(ciddMatrix).reshape(self.nEnergies, -1)
        self.nEnergySamples = int(self.ciddMatrix.shape[1])
        self.densityScaleFact = float(np.float32(densityScaleFact))
        self.densityVector = abi.f32(densityVector)
        self.nDensitySamples = int(self.densityVector.size)
        self.spScaleFact = float(np.float32(spScaleFact))
        self.spVector = abi.f32(spVector)
        self.nSpSamples = int(self.spVector.size)
        self.rRlScaleFact = float(np.float32(rRlScaleFact))
        self.rRlVector = abi.f32(rRlVector)
        self.nRRlSamples = int(self.rRlVector.size)
        # NUCLEAR_CORR tables (energy_struct.h:33-36), same shape as the cumulative IDD matrix; None when not loaded
        self.nucWeightMatrix = None if nucWeightMatrix is None else abi.f32(nucWeightMatrix).reshape(self.nEnergies, -1)
        self.nucSqSigmaMatrix = None if nucSqSigmaMatrix is None else abi.f32(nucSqSigmaMatrix).reshape(self.nEnergies, -1)

    def as_abi(self):
        """rtd_luts view of the arrays (keeps self alive while in use)."""
        s = abi.RtdLuts()
        s.n_energy_samples = self.nEnergySamples
        s.n_energies = self.nEnergies
        s.energies_per_u = abi.fptr(self.energiesPerU)
        s.peak_depths = abi.fptr(self.peakDepths)
        s.scale_facts = abi.fptr(self.scaleFacts)
        s.cidd_matrix = abi.fptr(self.ciddMatrix)
        s.n_density_samples = self.nDensitySamples
        s.density_scale_fact = self.densityScaleFact
        s.density_vector = abi.fptr(self.densityVector)
        s.n_sp_samples = self.nSpSamples
        s.sp_scale_fact = self.spScaleFact
        s.sp_vector = abi.fptr(self.spVector)
        s.n_rrl_samples = self.nRRlSamples
        s.rrl_scale_fact = self.rRlScaleFact
        s.rrl_vector = abi.fptr(self.rRlVector)
        if self.nucWeightMatrix is not None:
            s.nuc_weight_matrix = abi.fptr(self.nucWeightMatrix)
            s.nuc_sq_sigma_matrix = abi.fptr(self.nucSqSigmaMatrix)
        return s


def _read_tokens(path):
    with open(path) as fh:
        return fh.read().split()


NUC_FILES = {1: "nuclear_weights_and_sigmas_Soukup.txt", 2: "nuclear_weights_and_sigmas_Fluka.txt", 3: "nuclear_weights_and_sigmas_fit.txt"}


def read_lut_dir(directory, water_cube_test=False, nuclear_corr=0):
    """energyReader(dataPath) (energy_reader.cpp:12-101). water_cube_test selects the *_inc_water file (:77-93); nuclear_corr
    (abi.RTD_NUC_*) also reads the variant's nuclear table and checks its axes against the IDD table (:103-162)."""
    d = directory if directory.endswith("/") else directory + "/"
    t = _read_tokens(d + FILES["cidd"])
    nS, nE = int(t[0]), int(t[1])
    v = np.array(t[2:2 + 3 * nE + nS * nE], dtype=np.float32)
    e, p, s, m = v[:nE], v[nE:2 * nE], v[2 * nE:3 * nE], v[3 * nE:]

    def one(name):
        tt = _read_tokens(d + name)
        n = int(tt[0])
        return np.float32(tt[1]), np.array(tt[2:2 + n], dtype=np.float32)

    ds, dv = one(FILES["density"])
    ss, sv = one(FILES["sp"])
    rs, rv = one(FILES["rrl_water"] if water_cube_test else FILES["rrl"])
    nw = nq = None
    if nuclear_corr:
        name = NUC_FILES[int(nuclear_corr)]
        tt = _read_tokens(d + name)
        if int(tt[0]) != nS or int(tt[1]) != nE:
            raise RuntimeError("Number of samples or energies in %s different from proton_cumul_ddd_data.txt" % name)
        w = np.array(tt[2:2 + 3 * nE + 2 * nS * nE], dtype=np.float32)
        for k, (ref, what) in enumerate(((e, "Energies"), (p, "Peak depths"), (s, "Scale facts"))):
            if (np.abs(ref - w[k * nE:(k + 1) * nE]) > 0.01).any():
                raise RuntimeError("%s in %s different from proton_cumul_ddd_data.txt" % (what, name))
        nw, nq = w[3 * nE:3 * nE + nS * nE].reshape(nE, nS), w[3 * nE + nS * nE:].reshape(nE, nS)
    return EnergyStruct(e, p, s, m.reshape(nE, nS), ds, dv, ss, sv, rs, rv, nw, nq)


def write_lut_dir(directory, es, also_water=True):
    """Write an EnergyStruct in the reference's text layout."""
    os.makedirs(directory, exist_ok=True)
    d = directory if directory.endswith("/") else directory + "/"

    def row(a):
        return " ".join(repr(float(x)) for x in np.asarray(a, dtype=np.float32))

    with open(d + FILES["cidd"], "w") as fh:
        fh.write("%d %d\n\n" % (es.nEnergySamples, es.nEnergies))
        fh.write(row(es.energiesPerU) + "\n\n" + row(es.peakDepths) + "\n\n" + row(es.scaleFacts) + "\n\n")
        for r in es.ciddMatrix:
            fh.write(row(r) + "\n")

    def one(name, scale, vec):
        with open(d + name, "w") as fh:
            fh.write("%d %s\n\n" % (vec.size, repr(float(np.float32(scale)))))
            fh.write(row(vec) + "\n")

    one(FILES["density"], es.densityScaleFact, es.densityVector)
    one(FILES["sp"], es.spScaleFact, es.spVector)
    one(FILES["rrl"], es.rRlScaleFact, es.rRlVector)
    if also_water:
        one(FILES["rrl_water"], es.rRlScaleFact, es.rRlVector)
    if es.nucWeightMatrix is not None:
        for name in NUC_FILES.values():
            with open(d + name, "w") as fh:
                fh.write("%d %d\n\n" % (es.nEnergySamples, es.nEnergies))
                fh.write(row(es.energiesPerU) + "\n\n" + row(es.peakDepths) + "\n\n" + row(es.scaleFacts) + "\n\n")
                for r in es.nucWeightMatrix:
                    fh.write(row(r) + "\n")
                fh.write("\n")
                for r in es.nucSqSigmaMatrix:
                    fh.write(row(r) + "\n")


def _bragg_rows(energies, peaks, n_samples, peak_sample):
    """Cumulative integral depth dose rows: pristine curve (R-z)^(-0.435) blurred by range straggling.

    Row i is sampled at depth z_j = j / scale_i with scale_i = peak_sample / peak_i, is monotone
    non-decreasing, has its steepest rise (the Bragg peak of the differential curve) at sample
    peak_sample, and ends at 1.53e-7 * E_i like the measured tables.
    """
    rows = np.empty((len(energies), n_samples), dtype=np.float64)
    over = 8
    for i, (e, pk) in enumerate(zip(energies, peaks)):
        sigma = 0.012 * pk ** 0.935 + 0.3           # mm, range straggling + energy spread
        dz = pk / peak_sample / over
        z = (np.arange(n_samples * over + 1) + 0.5) * dz
        # place the pristine range a little behind the wanted peak position; refined below
        rng = pk + 0.6 * sigma
        for _ in range(6):
            zz = np.arange(-6 * sigma, 6 * sigma + dz, dz)
            g = np.exp(-0.5 * (zz / sigma) ** 2)
            g /= g.sum()
            t = np.clip(rng - z, 0.0, None)
            prist = np.where(t > 0, (t + 0.05) ** (-0.435) + 0.012 * (t + 0.05) ** 0.565, 0.0)
            h = (g.size - 1) // 2
            d = np.convolve(prist, g, mode="full")[h:h + prist.size]
            zpk = z[np.argmax(d)]
            rng += pk - zpk
        c = np.concatenate([[0.0], np.cumsum(d) * dz])
        rows[i] = c[::over][:n_samples]
        rows[i] *= 1.53e-7 * e / rows[i, -1]
    return rows


def synth_luts(n_energies=147, n_samples=1024, n_hu=3072, seed=0, nuclear=False):
    """Synthetic EnergyStruct with the shapes of the reference tables (1024x147, 3x3072). nuclear: also the two NUCLEAR_CORR
    tables — a halo fraction that grows with depth up to the peak (0..~12 %) and a squared halo sigma that grows with depth."""
    del seed  # tables are deterministic
    peaks = np.linspace(30.0, 320.0, n_energies)
    energies = (peaks / 0.01765) ** (1.0 / 1.8104)         # range-energy fit R = a E^p
    peak_sample = int(round(819.0 * n_samples / 1024.0))
    scale = peak_sample / peaks
    cidd = _bragg_rows(energies, peaks, n_samples, peak_sample)
    hu = np.arange(n_hu, dtype=np.float64)                  # HU + 1000
    # Schneider-like piecewise linear density: air 0.0012 -> water 1.0 at 1000 -> bone 2.24 at 3071
    dens = np.where(hu <= 1000, 0.0012 + (1.0 - 0.0012) * hu / 1000.0,
                    1.0 + (hu - 1000) * (2.24 - 1.0) / (n_hu - 1 - 1000))
    sp = np.where(hu <= 1000, 0.0011 + (1.0 - 0.0011) * hu / 1000.0,
                  1.0 + (hu - 1000) * (1.747 - 1.0) / (n_hu - 1 - 1000))
    rho = np.arange(n_hu, dtype=np.float64) / 1000.0        # density * 1000 index
    rrl = 0.00277 * (0.93 + 0.07 * np.clip(rho, 0, None)) + 0.0002 * np.clip(rho - 1.0, 0, None) ** 2
    nw = nq = None
    if nuclear:
        frac = np.arange(n_samples, dtype=np.float64)[None, :] / peak_sample           # depth / peak depth
        nw = 0.12 * np.clip(frac, 0.0, 1.0) * (peaks[:, None] / peaks.max()) ** 0.5
        nq = (0.023 * peaks[:, None] * np.clip(frac, 0.0, 1.3)) ** 2 + 1.0
    return EnergyStruct(energies, peaks, scale, cidd, 1.0, dens, 1.0, sp, 1000.0, rrl, nw, nq)
