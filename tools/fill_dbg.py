"""Clock stamps of k_fill's blocks on the bench field (RTD_FILL_DEBUG=1): per-CU load, by role / tile."""
import os, sys, collections
os.environ["RTD_FILL_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from raytracedicom_amd import engine, luts, scenarios
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "hetero"
es = luts.synth_luts()
if kind == "hetero":
    ct, _ = scenarios.hetero_phantom(n)
    scn = scenarios.hetero_ct(es, n=n, angles=[0.0], ct=ct)
else:
    scn = scenarios.water_cube(es, n=n)
eng = engine.Engine(0)
eng.set_luts(es); eng.set_ct(scn.ct)
d = eng.device_alloc(4 * scn.n_voxels); eng.device_zero(d, 4 * scn.n_voxels)
f = eng.create_field(scn.beams[0], scn.dims)
for i in range(3):
    f.compute(d); t, info = f.finish()
W, H, L = info["ray_dims"]
nT = (W // 32) * (H // 8)
dbg = f.fetch("fill_debug").reshape(-1, 4)
dur = (dbg[:, 1] - dbg[:, 0]).astype(float)
item = dbg[:, 3] >> 8; role = (dbg[:, 3] >> 4) & 1
tile = item % nT
hw = dbg[:, 2]; xcc = (hw >> 32) & 0xF; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (xcc << 16) | (se << 8) | (sh << 4) | cu
print("blocks", dbg.shape[0], "tiles", nT, "layers", L, " duration mean %.0f max %.0f (ticks)" % (dur.mean(), dur.max()))
for r in (0, 1):
    m = role == r
    print(" role %d: mean %.0f max %.0f" % (r, dur[m].mean(), dur[m].max()))
tx = W // 32
print(" mean duration by tile (rows = tile y):")
for ty in range(H // 8):
    print("   ", " ".join("%6.0f" % dur[tile == ty * tx + x].mean() for x in range(tx)))
load = collections.defaultdict(float); span = {}
for i in range(dbg.shape[0]):
    k = int(key[i]); load[k] += dur[i]
    a, b = span.get(k, (1 << 62, 0)); span[k] = (min(a, int(dbg[i, 0])), max(b, int(dbg[i, 1])))
ld = np.array(list(load.values())); sp = np.array([b - a for a, b in span.values()], float)
print("CUs %d: block-ticks per CU mean %.0f min %.0f max %.0f ; span per CU mean %.0f max %.0f" % (len(ld), ld.mean(), ld.min(), ld.max(), sp.mean(), sp.max()))
print("blocks per CU:", collections.Counter(collections.Counter(key.tolist()).values()))
f.destroy(); eng.close()
