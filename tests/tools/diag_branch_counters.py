#!/usr/bin/env python3
"""GPU diagnostic: per-branch auglag/TRON work counters of the HIP path vs the oracle.
usage: python scripts/diag_branch_counters.py [case] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
from oracle import oracle as O

case = sys.argv[1] if len(sys.argv) > 1 else "case1354pegase"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rho = {"case1354pegase": (1e1, 1e3), "case13659pegase": (5e1, 5e3), "case118": (4e2, 4e4)}[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
ref = O.OracleModel(g, rho[0], rho[1])
print("lib", pkg.LIB_PATH)
for it in range(1, iters + 1):
    t0 = time.perf_counter()
    norms, ms, msb = mod.run_fixed(1, first_major_iter=it, beta=1e3, time_branch=True)
    a, m, t = mod.branch_counters()
    ref.inner_prestep(); ref.update_x(it); ref.update_xbar(); ref.update_z(); ref.update_l(); r = ref.update_residual()
    if it <= 3 or it % 10 == 0:
        du = np.abs(mod.solution.u_curr - ref.sol["u_curr"]).max()
        print("it %3d branch %.3f ms | gpu auglag avg %.2f max %d, minor avg %.2f max %d, ticks max %d | oracle auglag %.2f | primres gpu %.6f ref %.6f | max|du| %.2e"
              % (it, msb, a.mean(), a.max(), m.mean(), m.max(), t.max(), ref.info.avg_auglag_it, norms[0], r[0], du))
        if a.max() >= 50:
            bad = np.flatnonzero(a >= 50)
            print("   branches at max_auglag:", len(bad), bad[:8], "mu:", mod.membuf[26, bad[:8]], "ref mu:", ref.membuf[bad[:8], 26])
