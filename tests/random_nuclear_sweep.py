"""Random sweep of the NUCLEAR_CORR restatement on a GPU box (not collected by pytest): seeded random scenarios, all three variants,
through tests/test_gpu_nuclear._run — radius classes and batch radii bit for bit, IDD, dose and gamma against the CPU oracle.
Usage: python tests/random_nuclear_sweep.py FIRST_SEED END_SEED."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.zeros(1, device="cuda")
from oracle import oracle as orc
from raytracedicom_amd import abi, engine, luts, scenarios
import test_gpu_nuclear as T
orc.lib(); orc.set_threads(16)
nuc = luts.synth_luts(nuclear=True)
n_ok = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([48, 64, 96]))
    variant = int(rng.choice([abi.RTD_NUC_SOUKUP, abi.RTD_NUC_FLUKA, abi.RTD_NUC_GAUSS_FIT]))
    spots = int(rng.integers(2, 9)); pitch = float(rng.choice([4.0, 6.0, 9.0])); n_layers = int(rng.choice([1, 2, 3, 5]))
    opt = abi.default_options(); opt.nuclear_corr = variant
    if rng.random() < 0.5:      # the reference's water cube: the tracer starts inside (entry step 0): the halo deposits
        scn = scenarios.water_cube(nuc, n=n, n_layers=n_layers, spots=spots, pitch=pitch, seed=int(rng.integers(1, 999)))
    else:                       # heterogeneous phantom, beam from air: primary scaled, halo cube empty
        ct, _ = scenarios.hetero_phantom(n, seed=int(rng.integers(1, 99)))
        beam = scenarios.make_field(nuc, n, 256.0 / n, (-128.0, -128.0, -106.0), float(rng.choice([0.0, 90.0, 180.0])), spots, pitch, n_layers,
                                    int(rng.integers(1, 999)), start_z=150.0)
        scn = scenarios.Scenario("hetero_air_gap", nuc, ct, (256.0 / n,) * 3, [beam])
    T._run(orc, engine, scn, opt)
    n_ok += 1
    print("seed", seed, "ok", n, variant, spots, pitch, n_layers, flush=True)
print("all", n_ok, "ok")
