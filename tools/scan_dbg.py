"""Clock stamps of k_trace_scan's blocks on the bench field (RTD_SCAN_DEBUG=1)."""
import os, sys
os.environ["RTD_SCAN_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from raytracedicom_amd import engine, luts, scenarios
es = luts.synth_luts()
ct, _ = scenarios.hetero_phantom(512)
scn = scenarios.hetero_ct(es, n=512, angles=[0.0], ct=ct)
eng = engine.Engine(0)
eng.set_luts(es); eng.set_ct(scn.ct)
d = eng.device_alloc(4 * scn.n_voxels); eng.device_zero(d, 4 * scn.n_voxels)
f = eng.create_field(scn.beams[0], scn.dims)
for i in range(3):
    f.compute(d); t, info = f.finish()
q = f.fetch("scan_debug").reshape(-1, 8).astype(np.float64)
t0 = q[:, 0].min()
# wave 0 (the WEPL chain) stamps: start, first chunk staged, then per 128-step chunk: walked (after the barrier), its own staging and stores done, past the end barrier
names = ["start", "chunk 0 staged", "chunk 0 walked", "chunk 0 stored", "barrier 0", "chunk 1 walked", "chunk 1 stored", "barrier 1"]
print("blocks", q.shape[0], "(shader clocks; stamps of a block's wave 0; the first two of four chunks)")
for i in range(1, 8):
    print("  phase -> %-16s mean %8.0f max %8.0f" % (names[i], (q[:, i] - q[:, i - 1]).mean(), (q[:, i] - q[:, i - 1]).max()))
f.destroy(); eng.close()
