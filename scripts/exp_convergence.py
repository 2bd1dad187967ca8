#!/usr/bin/env python3
"""GPU experiment: does the reference's two-level solve converge on a synthetic instance?
usage: exp_convergence.py CASE "dict(rho=(..), scale=.., grid=dict(..))" [max_outer] [time budget s]
Prints one line per outer iteration (the reference's own log columns) and the final status."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
case = sys.argv[1]
kw = eval(sys.argv[2]) if len(sys.argv) > 2 else {}
max_outer = int(sys.argv[3]) if len(sys.argv) > 3 else 20
budget = float(sys.argv[4]) if len(sys.argv) > 4 else 120.0
rho = kw.get("rho", {"case1354pegase": (1e1, 1e3), "case13659pegase": (5e1, 5e3), "case118": (4e2, 4e4),
                     "case_ACTIVSg70k": (3e4, 3e5), "synthetic1M": (5e1, 5e3)}[case])
g = syn.sized_like(case, seed=kw.get("seed", 0), **kw.get("grid", {}))
env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
par = env.params
par.scale = kw.get("scale", 1e-4)
mod.sync_params()
beta = par.initial_beta
sqrt_d = np.sqrt(mod.nvar)
cumul, t_all = 0, 0.0
print("== %s rho %s scale %g grid %s" % (case, rho, par.scale, kw.get("grid", {})), flush=True)
status = "IterationLimit"
for outer in range(1, max_outer + 1):
    pkg.admm_outer_prestep(env, mod)
    nzp = mod.info.norm_z_prev
    t0 = time.perf_counter()
    inner, res = mod.run_inner(outer, kw.get("inner", 1000), beta)
    dt = time.perf_counter() - t0
    t_all += dt
    cumul += inner
    a, m, t = mod.branch_counters()
    print("outer %2d inner %4d  %.3f ms/iter | primres %.4f dual %.4f normz %.4f mismatch %.4f tol %.4f beta %.3g | ticks avg %.2f p99 %d max %d auglag max %d obj %.1f"
          % (outer, inner, 1e3 * dt / inner, res[0], res[1], res[2], res[3], sqrt_d * par.outer_eps, beta, t.mean(), np.quantile(t, 0.99), t.max(), a.max(),
             0.0), flush=True)
    if res[3] <= sqrt_d * par.outer_eps:
        status = "Solved"
        break
    if t_all > budget:
        status = "budget"
        break
    par.beta = beta
    pkg.admm_update_lz(env, mod)
    if res[2] > par.theta * nzp:
        beta = min(par.inc_c * beta, 1e24)
pkg.admm_poststep(env, mod)
print("-> %s outer %d cumul %d time %.3f s  %.0f it/s objval %.4f" % (status, outer, cumul, t_all, cumul / t_all, mod.info.objval), flush=True)
