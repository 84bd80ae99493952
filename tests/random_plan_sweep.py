"""Random sweep of the multi-field machinery on a GPU box (not collected by pytest). Per seeded scenario — 2..5 beams at random gantry
angles, parallel or divergent, random incoming dose: (a) rtd_plan_compute on 1..3 handles of the one GPU, deferred CT, against
rtd_compute on one handle: bit-identical volumes; (b) the fused rtd_fields_transfer_init of all beams (every other one through its
exported message) against the sequence of rtd_field_transfer into a zeroed volume: bit-identical. Usage: FIRST_SEED END_SEED."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
torch.zeros(1, device="cuda")
from raytracedicom_amd import engine, luts, scenarios
synth = luts.synth_luts()
n_ok = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([48, 64, 96]))
    nb = int(rng.integers(2, 6))
    angles = [float(a) for a in rng.choice([0.0, 37.0, 45.0, 90.0, 141.0, 180.0, 225.0, 270.0, 300.0], size=nb, replace=True)]
    dist = (math.inf, math.inf) if rng.random() < 0.5 else (float(rng.uniform(900, 3000)), float(rng.uniform(900, 3000)))
    ct, _ = scenarios.hetero_phantom(n, seed=int(rng.integers(1, 99)))
    scn = scenarios.hetero_ct(synth, n=n, spots=int(rng.integers(2, 8)), pitch=float(rng.choice([4.0, 6.0, 9.0])), n_layers=int(rng.choice([1, 3, 6])),
                              angles=angles, source_dist=dist, ct=ct)
    base = rng.random(scn.ct.shape, dtype=np.float32) if rng.random() < 0.7 else np.zeros_like(scn.ct)
    want = base.copy()
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts); eng.set_ct(scn.ct)
        eng.compute(scn.beams, want)
    k = int(rng.integers(1, 4))
    got = base.copy()
    with engine.Plan([0] * k) as pl:
        pl.set_luts(scn.luts); pl.set_ct(scn.ct, deferred=True)
        pl.compute(scn.beams, got)
    np.testing.assert_array_equal(got, want, err_msg="plan on %d handles" % k)
    # (b) fused transfer against the sequence
    nv = scn.n_voxels
    with engine.Engine(0) as eng:
        eng.set_luts(scn.luts); eng.set_ct(scn.ct)
        a, b = eng.device_alloc(4 * nv), eng.device_alloc(4 * nv)
        eng.device_zero(a, 4 * nv)
        eng.to_device(b, np.full_like(scn.ct, 2.5))
        own = [eng.create_field(bm, scn.dims) for bm in scn.beams]
        fields, msgs = [], []
        for i, f in enumerate(own):
            f.compute_bev(); f.transfer(a)
            if i % 2:
                info, nbytes = f.wait_plan()
                m = eng.device_alloc(nbytes); f.export_bev(m, nbytes)
                r = eng.create_field(scn.beams[i], scn.dims, remote=True); r.attach_bev(m)
                fields.append(r); msgs.append(m)
            else:
                fields.append(f)
            f.finish()
        eng.transfer_fields_init(fields, b)
        eng.sync()
        x, y = np.empty_like(scn.ct), np.empty_like(scn.ct)
        eng.to_host(x, a); eng.to_host(y, b)
        np.testing.assert_array_equal(y, x, err_msg="fused transfer")
        for f in fields + own:
            f.destroy()
        for p in msgs + [a, b]:
            eng.device_free(p)
    n_ok += 1
    print("seed", seed, "ok", n, nb, angles, k, flush=True)
print("all", n_ok, "ok")
