#!/usr/bin/env python3
"""Batch radii of the bench workload's field at a list of gantry angles: how many (layer, step, tile) cells have a batch radius
beyond k_superpose_sweep's reach (16), in which layers and at which steps. What the rest launch (k_superpose_mfma) has to do."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from raytracedicom_amd import abi, engine, luts, scenarios


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    angles = [float(a) for a in sys.argv[2:]] or [0.0, 90.0, 135.0, 180.0, 270.0, 315.0]
    torch.cuda.init()
    es = luts.synth_luts()
    ct, _ = scenarios.hetero_phantom(n)
    scn = scenarios.hetero_ct(es, n=n, angles=angles, ct=ct)
    eng = engine.Engine(0)
    eng.set_options(abi.default_options())
    eng.set_luts(es)
    eng.set_ct(scn.ct)
    d = eng.device_alloc(4 * scn.n_voxels)
    eng.device_zero(d, 4 * scn.n_voxels)
    for a, beam in zip(angles, scn.beams):
        f = eng.create_field(beam, scn.dims)
        f.compute_bev(); f.transfer_init(d)
        t, info = f.finish()
        W, H, L = info["ray_dims"]
        S = int(beam.tracerSteps)
        tiles = (W // 32) * (H // 8)
        tr = f.fetch("tile_radius").reshape(L, S, tiles)
        eff = f.fetch("eff_radius").reshape(L, 34)
        cls = np.where(tr <= 33, tr, 0).astype(np.int64)
        e = np.take_along_axis(eff[:, None, :].repeat(S, 1), cls, axis=2)
        e = np.where(tr <= 32, e, -1)
        big = e > 16
        print("angle %g: classified cells %d, big %d (%.2f %%), max eff %d" % (a, int((tr <= 32).sum()), int(big.sum()), 100.0 * big.sum() / max(1, (tr <= 32).sum()), int(e.max())))
        for l in range(L):
            if big[l].any():
                ks = np.nonzero(big[l].any(axis=1))[0]
                allk = np.nonzero((tr[l] <= 32).any(axis=1))[0]
                print("   layer %2d: big steps %d..%d (%d steps, %d cells) of layer steps %d..%d; eff map tail %s" %
                      (l, ks.min(), ks.max(), ks.size, int(big[l].sum()), allk.min(), allk.max(), eff[l, 10:33].tolist()))
        f.destroy()


main()
