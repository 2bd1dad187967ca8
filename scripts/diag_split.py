#!/usr/bin/env python3
"""GPU diagnostic: the two-level driver (run_inner + the outer steps, as exaadmm_solve_two_level makes them) with option heavy_split against
the same solve without it; the state compared bit for bit after every outer iteration."""
import ctypes as C
import math
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
L = pkg.library()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
case = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "tests", "golden", "case9.m")
split = int(sys.argv[2]) if len(sys.argv) > 2 else 8

def state(mod):
    s = mod.solution
    a, m, t = mod.branch_counters()
    return [s.u_curr.copy(), s.v_curr.copy(), s.z_curr.copy(), s.l_curr.copy(), np.ascontiguousarray(mod.membuf).copy(), a, m, t]

runs = []
for sp in (0, split):
    os.environ["EXAADMM_HEAVY_SPLIT"] = str(sp)
    os.environ["EXAADMM_HEAVY_SPLIT_PREV"] = "0"
    env = pkg.AdmmEnv(case, 4e2, 4e4, verbose=0)
    mod = pkg.ModelAcopf(env)
    par = env.params
    par.outer_eps = 2e-5
    mod.sync_params()
    tol = math.sqrt(mod.nvar) * par.outer_eps
    beta, nz_prev, out = par.initial_beta, math.inf, []
    for outer in range(1, 26):
        v = C.c_double(0.0)
        pkg._check(L.exaadmm_norm_z(mod.handle, C.byref(v)), "norm_z")
        nz_prev = v.value
        inner, res = mod.run_inner(outer, 1000, beta)
        out.append((inner, res.copy(), state(mod), mod.resident_form(), mod.loop_downgrades(), beta))
        if res[3] <= tol:
            break
        pkg._check(L.exaadmm_update_lz(mod.handle, beta), "update_lz")
        if res[2] > par.theta * nz_prev:
            beta = min(par.inc_c * beta, 1e24)
    runs.append(out)
    mod.close()
names = ["u", "v", "z", "l", "membuf", "auglag", "minor", "ticks"]
print("outer iterations: %d / %d, cumulative %d / %d" % (len(runs[0]), len(runs[1]), sum(a[0] for a in runs[0]), sum(a[0] for a in runs[1])))
for k, (a, b) in enumerate(zip(*runs)):
    bad = [n for n, x, y in zip(names, a[2], b[2]) if not np.array_equal(x, y)]
    print("outer %2d: inner %3d / %3d, beta %g / %g, norms equal %s, launches %d / %d heavy %d / %d, downgrades %d / %d%s" % (k + 1, a[0], b[0], a[5], b[5], np.array_equal(a[1], b[1]), a[3]["launches"], b[3]["launches"],
          a[3]["heavy"], b[3]["heavy"], a[4], b[4], "" if not bad else "  DIFFERENT: " + ", ".join(bad)))
    for n, x, y in zip(names, a[2], b[2]):
        if not np.array_equal(x, y):
            d = np.flatnonzero(np.asarray(x).ravel() != np.asarray(y).ravel())
            print("      %s differs in %d of %d entries, first at %s; max abs diff %.3e" % (n, d.size, np.asarray(x).size, d[:10].tolist(), np.abs(np.asarray(x, dtype=float).ravel() - np.asarray(y, dtype=float).ravel()).max()))
    if bad:
        break
