#!/usr/bin/env python3
"""Quick GPU sanity check of one library build (EXAADMM_HIP_LIB): case9 anchor + first
iterations of a synthetic grid against the oracle, with per-branch work counters."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
from oracle import oracle as O
print("lib", os.path.basename(pkg.LIB_PATH), flush=True)
case9 = os.path.join(ROOT, "tests", "golden", "case9.m")
t0 = time.time()
env, mod = pkg.solve_acopf(case9, outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5, verbose=0)
print("case9: %s outer %d cumul %d obj %.4f  (%.2fs)" % (mod.info.status, mod.info.outer, mod.info.cumul, mod.info.objval, time.time() - t0), flush=True)
case = sys.argv[1] if len(sys.argv) > 1 else "case1354pegase"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rho = {"case1354pegase": (1e1, 1e3), "case13659pegase": (5e1, 5e3), "case118": (4e2, 4e4)}[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
ref = O.OracleModel(g, rho[0], rho[1])
for it in range(1, iters + 1):
    norms, ms, msb = mod.run_fixed(1, first_major_iter=it, beta=1e3, time_branch=True)
    a, m, t = mod.branch_counters()
    ref.inner_prestep(); ref.update_x(it); ref.update_xbar(); ref.update_z(); ref.update_l(); r = ref.update_residual()
    du = np.abs(mod.solution.u_curr - ref.sol["u_curr"]).max()
    mu = mod.membuf[26]
    print("it %2d branch %9.3f ms | auglag avg %.2f max %d ticks max %d | oracle auglag %.2f | primres %.6f ref %.6f | max|du| %.1e | mu in [%g, %g]"
          % (it, msb, a.mean(), a.max(), t.max(), ref.info.avg_auglag_it, norms[0], r[0], du, mu.min(), mu.max()), flush=True)
