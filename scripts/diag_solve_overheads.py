#!/usr/bin/env python3
"""What the timer of admm_two_level (time_overall) contains beside the iterations: a whole solve on a FRESH handle (as
solve_acopf makes it) against the same solve repeated on the handle that has already run once (alternate buffers allocated,
graph instantiated).  usage: diag_solve_overheads.py [case]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
rho, scale = {"case1354pegase": ((1e1, 1e3), 1e-4), "case13659pegase": ((5e1, 5e3), 1e-4), "case_ACTIVSg70k": ((3e4, 3e5), 1e-5)}[case]
g = syn.sized_like(case, seed=0)
for rep in range(3):
    env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    env.params.scale = scale
    env.params.outer_iterlim = 20
    env.params.inner_iterlim = 1000
    out = []
    for again in range(3):
        if again:
            pkg._check(pkg.library().exaadmm_init_solution(mod.handle, env.initial_rho_pq, env.initial_rho_va), "init_solution")
        pkg.admm_two_level(env, mod)
        i = mod.info
        out.append("%s %d/%d %.2f ms (%.0f it/s)" % ("fresh handle" if not again else "same handle", i.outer, i.cumul, 1e3 * i.time_overall, i.cumul / i.time_overall))
    print(case, " | ".join(out), flush=True)
    mod.close()
