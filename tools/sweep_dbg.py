"""Clock stamps of k_superpose_sweep's blocks on the bench field (RTD_SWEEP_DEBUG=1)."""
import os, sys
os.environ["RTD_SWEEP_DEBUG"] = "1"
os.environ["RTD_SEPARATE_KS_PLAN"] = "1"      # (the stamps are indexed by the blocks of the launch without the plan block in front)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from raytracedicom_amd import engine, luts, scenarios
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
es = luts.synth_luts()
ct, _ = scenarios.hetero_phantom(n)
scn = scenarios.hetero_ct(es, n=n, angles=[0.0], ct=ct)
eng = engine.Engine(0)
eng.set_luts(es); eng.set_ct(scn.ct)
d = eng.device_alloc(4 * scn.n_voxels); eng.device_zero(d, 4 * scn.n_voxels)
opt = None
f = eng.create_field(scn.beams[0], scn.dims)
for i in range(3):
    f.compute(d); t, info = f.finish()
print(t)
raw = f.fetch("sweep_debug")
nb = raw.size // 72
dbg = raw[:8 * nb].reshape(-1, 8)
wav = raw[8 * nb:].reshape(nb, 16, 4)
sel = dbg[:, 0] != 0
live = dbg[sel]
wl = wav[sel][:, :8]
print("per wave: stage+barrier mean %.1f  gather mean %.1f  wave main mean %.1f max-over-waves mean %.1f (x100 ticks)" % (
    wl[:, :, 0].mean() / 100, wl[:, :, 1].mean() / 100, wl[:, :, 2].mean() / 100, wl[:, :, 2].max(axis=1).mean() / 100))
print("blocks", dbg.shape[0], "live", live.shape[0])
t0 = live[:, 0].min()
setup = live[:, 1] - live[:, 0]; main = live[:, 2] - live[:, 1]; hand = live[:, 3] - live[:, 2]; comb = live[:, 4] - live[:, 3]
tick = 1e-8   # s_memtime: 100 MHz
for name, a in (("setup", setup), ("main", main), ("handoff", hand), ("combine(last only)", comb[(live[:, 5] & 1) == 1])):
    print("%-20s mean %8.2f us  max %8.2f us  sum %10.1f us" % (name, a.mean() * tick * 1e6, a.max() * tick * 1e6, a.sum() * tick * 1e6))
print("span first start -> last end: %.1f us" % ((live[:, 4].max() - t0) * tick * 1e6))
k = (live[:, 5] >> 32); nl = live[:, 6]
print("layers per block: mean %.2f max %d" % (nl.mean(), nl.max()))
order = np.argsort(live[:, 0])
print("start times (us) of first 8 / last 8 blocks:", ((live[order[:8], 0] - t0) * tick * 1e6).round(1), ((live[order[-8:], 0] - t0) * tick * 1e6).round(1))
print("main per layer (us): mean %.2f" % ((main / np.maximum(nl, 1)).mean() * tick * 1e6))
dur = (live[:, 4] - live[:, 0]).astype(float)
st = (live[:, 0] - t0).astype(float); en = (live[:, 4] - t0).astype(float)
for lo in range(0, 220, 20):
    m = (k >= lo) & (k < lo + 20)
    if m.any():
        print("k %3d-%3d: blocks %3d  layers/block %.2f  duration mean %7.0f max %7.0f (x100 ticks)  start mean %8.0f end max %8.0f" % (
            lo, lo + 19, m.sum(), nl[m].mean(), dur[m].mean() / 100, dur[m].max() / 100, st[m].mean() / 100, en[m].max() / 100))
print("kernel span (x100 ticks):", en.max() / 100, " sum of durations / 512:", dur.sum() / 512 / 100)
# per-CU concurrency: blocks of the same (XCC, SE, SH, CU) that overlap in time (timestamps of one XCD are comparable)
hw = live[:, 7]
xcc = (hw >> 32) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
key = (xcc << 16) | (se << 8) | (sh << 4) | cu
import collections
byc = collections.defaultdict(list)
for i in range(live.shape[0]):
    byc[int(key[i])].append((int(live[i, 0]), int(live[i, 4])))
print("CUs used:", len(byc), " blocks per CU: min %d max %d" % (min(len(v) for v in byc.values()), max(len(v) for v in byc.values())))
mx = []
busy1 = busy2 = span = 0
for v in byc.values():
    ev = sorted([(a, 1) for a, b in v] + [(b, -1) for a, b in v])
    c = 0; m = 0; last = ev[0][0]
    for t, d in ev:
        if c == 1: busy1 += t - last
        if c >= 2: busy2 += t - last
        last = t
        c += d; m = max(m, c)
    mx.append(m); span += ev[-1][0] - ev[0][0]
print("max concurrent blocks on a CU:", collections.Counter(mx))
print("fraction of CU-time with 1 block %.3f, with >= 2 blocks %.3f, idle within the CU's own span %.3f" % (busy1 / span, busy2 / span, 1 - (busy1 + busy2) / span))
print("mean span per CU (x100 ticks): %.0f" % (span / len(byc) / 100))
perx = collections.defaultdict(list)
for kk, v in byc.items():
    perx[kk >> 16].append((min(a for a, b in v), max(b for a, b in v), sum(b - a for a, b in v)))
for x, v in sorted(perx.items()):
    t0x = min(a for a, b, c in v)
    ends = sorted((b - t0x) / 100 for a, b, c in v)
    starts = sorted((a - t0x) / 100 for a, b, c in v)
    print("XCD %d: CUs %d  first block start spread %.0f..%.0f  CU end times min %.0f median %.0f max %.0f  busy block-ticks per CU mean %.0f" % (
        x, len(v), starts[0], starts[-1], ends[0], ends[len(ends) // 2], ends[-1], sum(c for a, b, c in v) / len(v) / 100))
f.destroy(); eng.close()
