#!/usr/bin/env python3
"""GPU vs oracle over several seeds / sizes / longer runs than the test suite: fused two-level loop,
iteration counts and solution distance.  A robustness sweep for changes of the lane code."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
from oracle import oracle as O
bad = 0
for case, rho, inner, nseed in (("case118", (4e2, 4e4), 150, 6), ("case1354pegase", (1e1, 1e3), 120, 6),
                                ("case13659pegase", (5e1, 5e3), 60, 2)):
    for seed in range(nseed):
        g = syn.sized_like(case, seed=seed)
        kw = dict(outer_iterlim=3, inner_iterlim=inner, rho_pq=rho[0], rho_va=rho[1], scale=1e-4)
        env, mod = pkg.solve_acopf(case, grid=g, verbose=0, **kw)
        ref = O.solve_acopf(case, grid=g, **kw)
        du = np.abs(mod.solution.u_curr - ref.sol["u_curr"]).max() / max(1.0, np.abs(ref.sol["u_curr"]).max())
        ok = (mod.info.cumul == ref.info.cumul) and du <= 1e-6
        bad += (not ok)
        print("%-15s seed %d: cumul gpu %4d oracle %4d | rel |du| %.2e | primres %.6g vs %.6g %s" % (
            case, seed, mod.info.cumul, ref.info.cumul, du, mod.info.primres, ref.info.primres, "" if ok else "  <-- MISMATCH"), flush=True)
print("mismatches:", bad)
