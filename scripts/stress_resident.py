#!/usr/bin/env python3
"""Stress of the resident loop's hand-over protocol (kernels.h k_resident): the ordering between a wavefront's data and the counter /
flag that announces them is a hardware property (sc1 accesses + s_waitcnt), and a violation would be timing dependent.  Many grids
(sizes from 64 to ~50 000 branches, several seeds, both cost models) and many REPETITIONS of the same run: every result must be bitwise
the two-launch loop's (EXAADMM_NO_RESIDENT=1) and bitwise the same from repetition to repetition.
usage: stress_resident.py [repetitions=40]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def run(g, rho, two_launch, plan, long_inner=0):
    if two_launch:
        os.environ["EXAADMM_NO_RESIDENT"] = "1"
    else:
        os.environ.pop("EXAADMM_NO_RESIDENT", None)
    env = pkg.AdmmEnv("stress", rho[0], rho[1], verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    kind = mod.loop_kind()
    sig = []
    it = 1
    for K in plan:
        norms, _, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
        it += K
        sig.append(norms.tobytes())
    if long_inner:   # a convergence-tested loop that never converges: a SECOND resident launch, which starts with a z / l pass and has heavy wavefronts
        inner, res = mod.run_inner(1000000, long_inner, 1e3)
    else:
        inner, res = mod.run_inner(2, 90, 1e3)
    s = mod.solution
    a, m, t = mod.branch_counters()
    sig += [res.tobytes(), bytes([inner % 256]), s.u_curr.tobytes(), s.v_curr.tobytes(), s.z_curr.tobytes(), s.l_curr.tobytes(),
            np.ascontiguousarray(mod.membuf).tobytes(), a.tobytes(), m.tobytes(), t.tobytes()]
    mod.close()
    return kind, sig


t0 = time.time()
bad = 0
cases = [(2, 64, 40, 0, "pegase"), (6, 200, 120, 1, "pegase"), (40, 700, 400, 2, "market"), (260, 1991, 1354, 3, "pegase"), (500, 5000, 3000, 4, "pegase"),
         (1500, 12000, 8000, 5, "market"), (4092, 20467, 13659, 6, "pegase"), (6000, 40000, 26000, 7, "pegase"), (7000, 50000, 33000, 8, "market"),
         # round 4: more tiles of 64 branches than SIMDs -- up to two tiles per resident wavefront (k_resident_t<2>)
         (8000, 70000, 52000, 9, "pegase"), (10390, 88207, 70000, 10, "market"), (12000, 110000, 80000, 11, "pegase")]
for ngen, nline, nbus, seed, cost in cases:
    g = syn.synthetic_grid(ngen, nline, nbus, seed=seed, cost=cost)
    rho = (5e1, 5e3) if cost == "pegase" else (3e2, 3e3)
    plan = (1, 2, 7, 30, 40, 3, 20)   # (from the loop's 65th iteration on: launches with heavy wavefronts)
    k0, ref = run(g, rho, True, plan)
    k1, first = run(g, rho, False, plan)
    same = (ref == first)
    print("grid %6d branches (seed %d, %s): loop kinds %d / %d, resident == two-launch: %s" % (nline, seed, cost, k0, k1, same), flush=True)
    bad += (not same)
g = syn.sized_like("case13659pegase", seed=0)
k0, ref = run(g, (5e1, 5e3), True, (1, 2, 7, 30, 40, 3, 20))
nbad = 0
for r in range(reps):
    _, sig = run(g, (5e1, 5e3), False, (1, 2, 7, 30, 40, 3, 20))
    nbad += (sig != ref)
print("13 659 buses: %d repetitions of the resident run, %d differ from the two-launch result" % (reps, nbad), flush=True)
g = syn.sized_like("case1354pegase", seed=0)
k0, ref = run(g, (1e1, 1e3), True, (1, 2, 7, 30, 40, 3, 20))
nbad2 = 0
for r in range(reps):
    _, sig = run(g, (1e1, 1e3), False, (1, 2, 7, 30, 40, 3, 20))
    nbad2 += (sig != ref)
print("1 354 buses: %d repetitions of the resident run, %d differ from the two-launch result" % (reps, nbad2), flush=True)
repsL = max(1, reps // 10)
for case, rho in (("case1354pegase", (1e1, 1e3)), ("case13659pegase", (5e1, 5e3))):
    g = syn.sized_like(case, seed=0)
    k0, ref = run(g, rho, True, (70,), 1040)
    nb = 0
    for r in range(repsL):
        _, sig = run(g, rho, False, (70,), 1040)
        nb += (sig != ref)
    print("%s: %d repetitions of a 1 040-iteration inner loop (two resident launches), %d differ from the two-launch result" % (case, repsL, nb), flush=True)
    nbad2 += nb
g = syn.sized_like("case_ACTIVSg70k", seed=0)
k0, ref = run(g, (3e4, 3e5), True, (1, 2, 7, 30, 40, 3, 20))
nbad3 = 0
reps3 = max(1, reps // 5)
for r in range(reps3):
    _, sig = run(g, (3e4, 3e5), False, (1, 2, 7, 30, 40, 3, 20))
    nbad3 += (sig != ref)
print("88 207 branches (two tiles per wavefront): %d repetitions of the resident run, %d differ from the two-launch result" % (reps3, nbad3), flush=True)
print("TOTAL mismatches: %d (%.0f s)" % (bad + nbad + nbad2 + nbad3, time.time() - t0))
sys.exit(1 if (bad + nbad + nbad2 + nbad3) else 0)
