"""Clock stamps of k_superpose_uniform4's blocks on the reference's water cube C2 (RTD_UNIFORM_DEBUG=1)."""
import os, sys
os.environ["RTD_UNIFORM_DEBUG"] = "1"
os.environ.setdefault("RTD_UNIFORM_V4", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np
import torch
torch.cuda.init()
from raytracedicom_amd import engine, luts, scenarios
es = luts.synth_luts()
scn = scenarios.water_cube(es, n=256, n_layers=20)
eng = engine.Engine(0)
eng.set_luts(es); eng.set_ct(scn.ct)
d = eng.device_alloc(4 * scn.n_voxels); eng.device_zero(d, 4 * scn.n_voxels)
f = eng.create_field(scn.beams[0], scn.dims)
for i in range(3):
    f.compute(d); t, info = f.finish()
print(t)
raw = f.fetch("uniform_debug").reshape(-1, 16)
live = raw[raw[:, 0] != 0]
tick = 1.0   # (s_memtime counts shader clocks here: the figures below are cycles)
t0 = live[:, 0].min()
k = live[:, 5] >> 32; part = live[:, 5] & 0xFFFFFFFF; nA = live[:, 6]
print("blocks", raw.shape[0], "live", live.shape[0], " layers per block: mean %.1f max %d" % (nA.mean(), nA.max()))
for name, a in (("layer list", live[:, 1] - live[:, 0]), ("weight table", live[:, 2] - live[:, 1]), ("layer loop", live[:, 3] - live[:, 2]), ("store", live[:, 4] - live[:, 3])):
    print("%-14s mean %7.2f cyc  max %7.2f cyc" % (name, a.mean() * tick, a.max() * tick))
print("span first start -> last end: %.1f cyc;  last start at %.1f cyc" % ((live[:, 4].max() - t0) * tick, (live[:, 0].max() - t0) * tick))
loop = (live[:, 3] - live[:, 2]).astype(float)
print("layer loop per layer: mean %.2f cyc" % ((loop / np.maximum(nA, 1)).mean() * tick))
work = live[:, 8:12].astype(float); bar = live[:, 12:16].astype(float)
print("per wave, per layer: arithmetic (incl. fetch issue) mean %.2f cyc  max-over-waves mean %.2f cyc;  the two barriers + store: mean %.2f cyc" % (
    (work.mean(axis=1) / np.maximum(nA, 1)).mean() * tick, (work.max(axis=1) / np.maximum(nA, 1)).mean() * tick, (bar.mean(axis=1) / np.maximum(nA, 1)).mean() * tick))
for lo in range(0, int(k.max()) + 1, 20):
    m = (k >= lo) & (k < lo + 20)
    if m.any():
        print("k %3d-%3d: blocks %3d layers %5.1f  loop/layer %5.2f cyc  work/layer (wave mean) %5.2f  start mean %6.1f end max %6.1f cyc" % (
            lo, lo + 19, m.sum(), nA[m].mean(), (loop[m] / np.maximum(nA[m], 1)).mean() * tick, (work[m].mean(axis=1) / np.maximum(nA[m], 1)).mean() * tick,
            (live[m, 0] - t0).mean() * tick, (live[m, 4] - t0).max() * tick))
hw = live[:, 7]
xcc = (hw >> 32) & 0xF; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
key = (xcc << 16) | (se << 8) | (sh << 4) | cu
byc = collections.Counter(int(x) for x in key)
print("CUs cyced:", len(byc), " blocks per CU: min %d max %d" % (min(byc.values()), max(byc.values())))
