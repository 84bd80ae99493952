#!/usr/bin/env python3
"""The reference's own configuration C2 (WATER_CUBE_TEST: 256^3 water, 33x33 spots x 20 layers, 128x128 rays, main.cu:39-99) run
K times on one GPU with resident inputs, for a rocprofv3 --kernel-trace --stats pass (profiles/collect.sh -> profiles/r02_c2_*).
Prints one JSON line: ms per field (pipelined by one, like bench.py) and the per-stage device times."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from raytracedicom_amd import abi, engine, luts, scenarios


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.cuda.init()
    es = luts.synth_luts()
    scn = scenarios.water_cube(es, n=256, n_layers=20)
    eng = engine.Engine(0)
    opt = abi.default_options()
    opt.fine_grained_timing = 1
    eng.set_options(opt)
    eng.set_luts(es)
    eng.set_ct(scn.ct)
    n = scn.n_voxels
    d = eng.device_alloc(4 * n)
    eng.device_zero(d, 4 * n)
    flds = [eng.create_field(scn.beams[0], scn.dims) for _ in range(2)]
    for f in flds:
        f.compute_bev(); f.transfer_init(d); f.finish()
    eng.sync()
    acc = {}
    t0 = time.perf_counter()
    pending = None
    for i in range(steps):
        f = flds[i % 2]
        f.compute_bev()
        f.transfer_init(d)
        if pending is not None:
            t, info = pending.finish()
            for k, v in t.items():
                if isinstance(v, float):
                    acc[k] = acc.get(k, 0.0) + v
        pending = f
    t, info = pending.finish()
    for k, v in t.items():
        if isinstance(v, float):
            acc[k] = acc.get(k, 0.0) + v
    eng.sync()
    el = time.perf_counter() - t0
    W, H, L = info["ray_dims"]
    R, P, SA = W * H, (W + 64) * (H + 64), info["live_steps"]
    ks_bytes = 8 * R * SA + 8 * P * SA
    ks_ms = acc["superp_kernel_ms"] / steps
    print(json.dumps({"workload": "C2: reference water cube 256^3, 33x33 spots x 20 layers, ray grid %dx%d, live steps %d, max radius %d"
                                  % (W, H, SA, info["max_radius"]),
                      "ms_per_field": round(1000 * el / steps, 4), "stage_ms": {k: round(v / steps, 4) for k, v in acc.items()},
                      "superposition_algorithmic_bytes": ks_bytes, "superposition_gbs": round(ks_bytes / (ks_ms * 1e-3) / 1e9, 1),
                      "superposition_frac_of_8TBs": round(ks_bytes / (ks_ms * 1e-3) / 1e9 / 8000.0, 4)}))
    for f in flds:
        f.destroy()
    eng.device_free(d)
    eng.close()


if __name__ == "__main__":
    main()
