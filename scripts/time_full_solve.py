#!/usr/bin/env python3
"""GPU: wall time of fused two-level solves (the reference's metric: cumul / time_overall)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
case9 = os.path.join(ROOT, "tests", "golden", "case9.m")
for rep in range(2):
    env, mod = pkg.solve_acopf(case9, outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5, verbose=0)
    i = mod.info
    print("case9: %s outer %d cumul %d  time %.4f s  %.0f it/s" % (i.status, i.outer, i.cumul, i.time_overall, i.cumul / i.time_overall))
for case, rho in (("case1354pegase", (1e1, 1e3)), ("case13659pegase", (5e1, 5e3))):
    g = syn.sized_like(case, seed=0)
    for rep in range(2):
        env, mod = pkg.solve_acopf(case, grid=g, outer_iterlim=4, inner_iterlim=1000, rho_pq=rho[0], rho_va=rho[1], verbose=0)
        i = mod.info
        print("%s: outer %d cumul %d  time %.4f s  %.0f it/s  primres %.10g" % (case, i.outer, i.cumul, i.time_overall, i.cumul / i.time_overall, i.primres))
