#!/usr/bin/env python3
"""Whole-box transfer of one field of the bench workload at 0 / 90 / 45 degrees (gather layouts: lanes along x, along z, along x with
an oblique box), 20 back-to-back launches each: rtd_field_transfer_init, rtd_field_transfer (read-modify-write) and the fused
rtd_fields_transfer_init with that one field. Microseconds per launch (profiles/r02_transfer_probe.txt)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracedicom_amd import abi, engine, luts, scenarios
n = 512
es = luts.synth_luts()
ct_np, _ = scenarios.hetero_phantom(n)
angles = [0.0, 90.0, 45.0]
scn = scenarios.hetero_ct(es, n=n, angles=angles, ct=ct_np)
eng = engine.Engine(0)
opt = abi.default_options(); opt.fine_grained_timing = 1
eng.set_options(opt); eng.set_luts(es); eng.set_ct(scn.ct)
d = eng.device_alloc(4 * scn.n_voxels); eng.device_zero(d, 4 * scn.n_voxels)
for a, beam in zip(angles, scn.beams):
    f = eng.create_field(beam, scn.dims)
    f.compute_bev(); f.transfer_init(d); t, info = f.finish()
    lo, hi = info["dose_box_min"], info["dose_box_max"]
    res = {}
    for name, fn in (("init", lambda: f.transfer_init(d)), ("rmw", lambda: f.transfer(d)), ("fused", lambda: eng.transfer_fields_init([f], d, lo, hi))):
        fn(); eng.sync()
        t0 = time.perf_counter()
        for i in range(20):
            fn()
        eng.sync()
        res[name] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
    print(a, "box voxels", t["transfer_voxels"], res, flush=True)
    f.destroy()
