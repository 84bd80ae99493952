#!/usr/bin/env python3
"""Per-block clock stamps of k_superpose_sweep_big (RTD_SWEEP_DEBUG=1): where a block of the second sweep launch spends its time."""
import os
import sys

import numpy as np

os.environ["RTD_SWEEP_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from raytracedicom_amd import abi, engine, luts, scenarios


def main():
    angle = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
    n = 512
    torch.cuda.init()
    es = luts.synth_luts()
    ct, _ = scenarios.hetero_phantom(n)
    scn = scenarios.hetero_ct(es, n=n, angles=[angle], ct=ct)
    eng = engine.Engine(0)
    eng.set_options(abi.default_options())
    eng.set_luts(es)
    eng.set_ct(scn.ct)
    d = eng.device_alloc(4 * scn.n_voxels)
    eng.device_zero(d, 4 * scn.n_voxels)
    f = eng.create_field(scn.beams[0], scn.dims)
    for _ in range(3):
        f.compute_bev(); f.transfer_init(d)
        t, info = f.finish()
    q = f.fetch("sweep_big_debug").reshape(-1, 48)
    live = q[q[:, 0] != 0]
    t0 = live[:, 0].min()
    tk = 100.0   # ticks per unit
    print("blocks", q.shape[0], "live", live.shape[0], "last", int((live[:, 5] & 1).sum()))
    for name, a, b in (("setup", 0, 1), ("rows", 1, 2), ("handoff", 2, 3), ("tail(last: combine)", 3, 4), ("total", 0, 4)):
        dlt = (live[:, b] - live[:, a]) / tk
        print("%-22s mean %9.1f  max %9.1f   (x100 ticks)" % (name, dlt.mean(), dlt.max()))
    lastm = (live[:, 5] & 1) == 1
    print("combine of last blocks: mean %.1f max %.1f" % (((live[lastm, 4] - live[lastm, 3]) / tk).mean(), ((live[lastm, 4] - live[lastm, 3]) / tk).max()))
    print("start spread %.1f  end max %.1f" % ((live[:, 0].max() - t0) / tk, (live[:, 4].max() - t0) / tk))
    big = live[:, 8:16]
    print("big row-layers per wave: mean %.1f max %d; per block sum mean %.1f max %d; layers per block mean %.1f" % (big.mean(), big.max(), big.sum(1).mean(), big.sum(1).max(), live[:, 6].mean()))
    pw = live[:, 16:48].reshape(-1, 8, 4) / tk
    print("per wave (x100 ticks): build mean %.1f  multiply mean %.1f  flush+wait mean %.1f  wave span mean %.1f max %.1f" %
          (pw[:, :, 0].mean(), pw[:, :, 1].mean(), pw[:, :, 2].mean(), pw[:, :, 3].mean(), pw[:, :, 3].max()))
    nb = np.maximum(big, 1)
    print("per big row-layer (ticks): build %.0f multiply %.0f" % ((pw[:, :, 0] * tk / nb).mean(), (pw[:, :, 1] * tk / nb).mean()))
    rows = (live[:, 2] - live[:, 1]) / tk
    i = np.argsort(-rows)[:5]
    for j in i:
        print("  long block: k %d g %d rows-phase %.1f big/wave %s" % (live[j, 5] >> 32, (live[j, 5] >> 8) & 255, rows[j], big[j].tolist()))


main()
