#!/usr/bin/env python3
"""Stress of the ADOPTION inside resident launches (kernels.h ResidentHeavy): long fixed-count launches with heavy wavefronts, adoption thresholds 3 and 4,
many repetitions -- every result bitwise the two-launch loop's.  usage: stress_adoption.py [repetitions=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def run(case, g, rho, two_launch, adopt, plan):
    os.environ.pop("EXAADMM_NO_RESIDENT", None)
    if two_launch:
        os.environ["EXAADMM_NO_RESIDENT"] = "1"
    os.environ["EXAADMM_HEAVY_ADOPT"] = str(adopt)
    env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    sig, it = [], 1
    for K in plan:
        norms, _, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
        it += K
        sig.append(norms.tobytes())
    s = mod.solution
    a, m, t = mod.branch_counters()
    sig += [s.u_curr.tobytes(), s.v_curr.tobytes(), s.z_curr.tobytes(), s.l_curr.tobytes(), np.ascontiguousarray(mod.membuf).tobytes(), a.tobytes(), m.tobytes(), t.tobytes()]
    dg = mod.loop_downgrades()
    mod.close()
    return sig, dg


t0 = time.time()
total = 0
for case, rho in (("case13659pegase", (5e1, 5e3)), ("case1354pegase", (1e1, 1e3))):
    g = syn.sized_like(case, seed=0)
    plan = (70, 200, 37, 120)
    ref, _ = run(case, g, rho, True, 4, plan)
    for adopt in (4, 3):
        bad = down = 0
        for r in range(reps):
            sig, dg = run(case, g, rho, False, adopt, plan)
            bad += (sig != ref)
            down += dg
        print("%s, adoption from %d trips: %d repetitions of %s iterations, %d differ from the two-launch result, %d downgrades" % (case, adopt, reps, "+".join(map(str, plan)), bad, down), flush=True)
        total += bad + down
print("TOTAL: %d (%.0f s)" % (total, time.time() - t0))
sys.exit(1 if total else 0)
