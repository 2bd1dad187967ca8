#!/usr/bin/env python3
"""GPU diagnostic: who are the stragglers of the branch launch in the steady state?  Per-branch counters (augmented-
Lagrangian iterations, TRON minor iterations, solver-loop trips) after N iterations of the benchmark state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rho = {"case1354pegase": (1e1, 1e3), "case13659pegase": (5e1, 5e3), "case_ACTIVSg70k": (3e4, 3e5), "synthetic1M": (5e1, 5e3)}[case]
g = syn.sized_like(case, seed=0, **(dict(cost="market") if case == "case_ACTIVSg70k" else {}))
env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
env.params.scale = 1e-5 if case == "case_ACTIVSg70k" else 1e-4
mod.sync_params()
mod.run_fixed(n, first_major_iter=1, beta=1e3)
for rep in range(3):
    norms, ms, msb = mod.run_fixed(1, first_major_iter=n + 1 + rep, beta=1e3, time_branch=True)
    a, m, t = mod.branch_counters()
    print("iteration %d: k_branch %.4f ms | trips: %s" % (n + 1 + rep, msb, " ".join("%d:%d" % (k, (t == k).sum()) for k in range(1, t.max() + 1) if (t == k).sum())))
    top = np.argsort(-t)[:8]
    print("   heaviest branches (trips, auglag its, minor its total): " + ", ".join("%d:(%d,%d,%d)" % (b, t[b], a[b], m[b]) for b in top))
    mu = mod.membuf[26]
    print("   penalty of the heaviest: %s ; rateA-limited? %s" % (mu[top], (np.asarray(g["rateA"])[top] < 999)))
