#!/usr/bin/env python3
"""GPU diagnostic: which branches make some inner iterations slow? (outer 6 of the synthetic headline grid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
g = syn.sized_like("case13659pegase", seed=0)
env = pkg.AdmmEnv("syn", 5e1, 5e3, verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
par = env.params
beta = par.initial_beta
for outer in range(1, 6):
    pkg.admm_outer_prestep(env, mod)
    nzp = mod.info.norm_z_prev
    inner, res = mod.run_inner(outer, 1000, beta)
    par.beta = beta
    pkg.admm_update_lz(env, mod)
    if res[2] > par.theta * nzp:
        beta = min(par.inc_c * beta, 1e24)
print("entering outer 6, beta", beta, flush=True)
worst = []
for it in range(1, 1001):
    t0 = time.perf_counter()
    norms, ms, msb = mod.run_fixed(1, first_major_iter=it, beta=beta, time_branch=True)
    if msb > 1.0:
        a, m, t = mod.branch_counters()
        b = int(np.argmax(t))
        worst.append((it, msb, b, int(a[b]), int(m[b]), int(t[b])))
        if len(worst) <= 12:
            bb = g["brBusIdx"].reshape(-1, 2)[b]
            print("it %4d k_branch %.2f ms: branch %d (buses %d-%d) auglag %d minor %d ticks %d | rateA %.3g YffR %.3g YffI %.3g tap? %s | mu %g | #ticks>100: %d"
                  % (it, msb, b, bb[0], bb[1], a[b], m[b], t[b], g["rateA"][b], g["YffR"][b], g["YffI"][b], abs(g["YffR"][b] - g["YttR"][b]) > 1e-12,
                     mod.membuf[26, b], (t > 100).sum()), flush=True)
print("slow iterations (>1 ms):", len(worst), "distinct branches:", len({w[2] for w in worst}))
