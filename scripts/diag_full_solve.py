#!/usr/bin/env python3
"""GPU diagnostic: how does the cost of an inner iteration evolve over a full two-level solve?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
rho = {"case1354pegase": (1e1, 1e3), "case13659pegase": (5e1, 5e3), "case118": (4e2, 4e4)}[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
par = env.params
beta = par.initial_beta
sqrt_d = np.sqrt(mod.nvar)
cumul = 0
for outer in range(1, int(sys.argv[2]) + 1 if len(sys.argv) > 2 else 9):
    pkg.admm_outer_prestep(env, mod)
    nzp = mod.info.norm_z_prev
    t0 = time.perf_counter()
    inner, res = mod.run_inner(outer, 1000, beta)
    dt = time.perf_counter() - t0
    cumul += inner
    a, m, t = mod.branch_counters()
    print("outer %2d inner %4d  %.3f ms/iter | primres %.4f mismatch %.4f beta %.3g | last x-update: auglag avg %.2f max %d, ticks avg %.1f p99 %d max %d, #branches at max_auglag %d"
          % (outer, inner, 1e3 * dt / inner, res[0], res[3], beta, a.mean(), a.max(), t.mean(), np.quantile(t, 0.99), t.max(), (a >= 50).sum()), flush=True)
    if res[3] <= sqrt_d * par.outer_eps:
        print("Solved")
        break
    par.beta = beta
    pkg.admm_update_lz(env, mod)
    if res[2] > par.theta * nzp:
        beta = min(par.inc_c * beta, 1e24)
