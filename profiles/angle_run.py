#!/usr/bin/env python3
"""The bench workload's field at the gantry angles bench.py --gpus N assigns to its ranks (r*360/N), each run K times alone on one
GPU with resident inputs: per-stage device times per angle. What the N-GPU step time is made of: every rank computes its own field
up to the BEV dose and transfers every field clipped to its slab, so the slowest angle sets the pace (profiles/r02_angles.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from raytracedicom_amd import abi, engine, luts, scenarios


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    torch.cuda.init()
    es = luts.synth_luts()
    ct, _ = scenarios.hetero_phantom(n)
    angles = [float(a) for a in sys.argv[3:]] or [0.0, 45.0, 90.0, 135.0, 180.0, 225.0, 270.0, 315.0]
    scn = scenarios.hetero_ct(es, n=n, angles=angles, ct=ct)
    eng = engine.Engine(0)
    opt = abi.default_options()
    opt.fine_grained_timing = 1
    eng.set_options(opt)
    eng.set_luts(es)
    eng.set_ct(scn.ct)
    d = eng.device_alloc(4 * scn.n_voxels)
    eng.device_zero(d, 4 * scn.n_voxels)
    out = {}
    for a, beam in zip(angles, scn.beams):
        f = eng.create_field(beam, scn.dims)
        acc = {}
        for i in range(steps + 2):
            f.compute_bev(); f.transfer_init(d)
            t, info = f.finish()
            if i >= 2:
                for k, v in t.items():
                    if isinstance(v, float):
                        acc[k] = acc.get(k, 0.0) + v
        row = {k: round(v / steps, 4) for k, v in acc.items()}
        row.update(ray_dims=info["ray_dims"], steps=int(beam.tracerSteps), live_steps=info["live_steps"], max_radius=info.get("max_radius"),
                   box_voxels=int(t.get("transfer_voxels", 0)))
        out["%g" % a] = row
        print(a, json.dumps(row), flush=True)
        f.destroy()
    print(json.dumps(out))


main()
