# (not collected by pytest: an analysis script on the CPU oracle, kept beside the tests because only tests may use the oracle)
"""Statistics of the superposition's work on a bench field (CPU oracle): per (layer, step) the dose-carrying rectangle, the number of
dose-carrying rays, the batch radii; per step the max radius."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracedicom_amd import luts, scenarios
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "hetero"
deg = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
es = luts.synth_luts()
if kind == "hetero":
    ct, _ = scenarios.hetero_phantom(n)
    scn = scenarios.hetero_ct(es, n=n, angles=[deg], ct=ct)
else:
    scn = scenarios.water_cube(es, n=n)
beam = scn.beams[0]
d = np.zeros_like(scn.ct)
t = time.time()
of = oracle.run_field(scn, beam, d, keep_layers=True)
print("oracle %.1fs" % (time.time() - t), of.info)
W, H, L = of.info["ray_dims"]; S = beam.tracerSteps
idd = of.get("idd").reshape(L, S, H, W)
tr = of.get("tile_radius").reshape(L, S, H // 8, W // 32)
eff = of.get("eff_radius").reshape(L, -1)
plan = of.get("layer_plan").reshape(L, 8)
first = of.info["beam_first_inside"]; calc = of.info["beam_first_calculated_passive"]
tot_src = 0; tot_rect = 0; n_ls = 0
rho_hist = np.zeros(40, np.int64)         # sources by own batch radius
macs = 0
per_step_maxrho = np.zeros(S, int)
wid = []; hei = []
for l in range(L):
    lfp = int(plan[l, 6])
    for k in range(first, lfp):
        m = idd[l, k] > 0
        c = int(m.sum())
        if c == 0: continue
        n_ls += 1
        ys, xs = np.nonzero(m)
        w, h = xs.max() - xs.min() + 1, ys.max() - ys.min() + 1
        wid.append(w); hei.append(h)
        tot_src += c; tot_rect += w * h
        r = tr[l, k]
        rr = np.where(r <= 32, eff[l][np.minimum(r, 33)], -1)
        rmap = np.repeat(np.repeat(rr, 8, axis=0), 32, axis=1)
        rs = rmap[m]
        rho_hist += np.bincount(rs[rs >= 0], minlength=40)[:40]
        macs += int(((2 * rs[rs >= 0] + 1) ** 2).sum())
        per_step_maxrho[k] = max(per_step_maxrho[k], rs.max())
print("live (layer,step):", n_ls, "sources:", tot_src, "per slice:", tot_src / n_ls, "rect area per slice:", tot_rect / n_ls,
      "mean w,h:", np.mean(wid), np.mean(hei), "max w,h", max(wid), max(hei))
print("useful MACs: %.3g" % macs)
print("sources by batch radius:", {i: int(v) for i, v in enumerate(rho_hist) if v})
print("steps by max radius:", {int(i): int(v) for i, v in zip(*np.unique(per_step_maxrho[first:calc], return_counts=True))})
cum = np.cumsum(rho_hist) / rho_hist.sum()
print("cum frac by radius:", {i: round(float(c), 3) for i, c in enumerate(cum) if rho_hist[i]})
