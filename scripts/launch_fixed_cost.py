#!/usr/bin/env python3
"""GPU diagnostic: device time of ONE fixed-count run of K iterations from one state, K = 5 ... 400 (exaadmm_run_fixed's own HIP-event
bracket): T(K) = a + b K.  For the resident loop `a` is what a launch costs before / after its iterations; the two-launch loop
(EXAADMM_NO_RESIDENT=1 / EXAADMM_RESIDENT_TILES=0) is the comparison.   usage: launch_fixed_cost.py [case]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, exaadmm_loader
import bench as B
pkg = exaadmm_loader.load_package(); syn = exaadmm_loader.load_submodule("synthetic")
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
par = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv(case, par["rho_pq"], par["rho_va"], tight_factor=0.99, verbose=0, grid=g); mod = pkg.ModelAcopf(env)
env.params.scale = par["scale"]; mod.sync_params()
mod.run_fixed(320, first_major_iter=1, beta=1e3)
mod.state_save()
Ks = [5, 10, 20, 50, 100, 200, 400]
rows = []
for K in Ks:
    ts = []
    for rep in range(5):
        mod.state_restore()
        _, ms, _ = mod.run_fixed(K, first_major_iter=321, beta=1e3)
        ts.append(ms)
    rows.append((K, sorted(ts)[2]))
A = np.array([[1.0, k] for k, _ in rows]); y = np.array([t for _, t in rows])
a, b = np.linalg.lstsq(A, y, rcond=None)[0]
print("# %s loop kind %d: median device ms of a run of K iterations from the state after 320 iterations" % (case, mod.loop_kind()))
for k, t in rows:
    print("  K = %4d: %8.3f ms  (%.2f us per iteration)" % (k, t, 1e3 * t / k))
print("  fit: %.3f ms + %.2f us x K" % (a, 1e3 * b))
