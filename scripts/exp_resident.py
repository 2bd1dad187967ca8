#!/usr/bin/env python3
"""Resident loop experiment (kernels.h k_resident, EXAADMM_RESIDENT=1): run_fixed(K) in ONE launch against the default two launches per
iteration -- bitwise comparison of the state after K iterations, and ms per iteration.  usage: exp_resident.py [case ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
RHO = {"case1354pegase": ((1e1, 1e3), 1e-4), "case13659pegase": ((5e1, 5e3), 1e-4)}


def state(mod):
    s = mod.solution
    a, m, t = mod.branch_counters()
    return [s.u_curr.copy(), s.v_curr.copy(), s.z_curr.copy(), s.l_curr.copy(), np.ascontiguousarray(mod.membuf).copy(), a, m, t]


for case in (sys.argv[1:] or ["case1354pegase", "case13659pegase"]):
    g = syn.sized_like(case, seed=0)
    rho, scale = RHO[case]
    outs = {}
    for res in ("0", "1"):
        os.environ["EXAADMM_RESIDENT"] = res
        env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
        mod = pkg.ModelAcopf(env)
        env.params.scale = scale
        mod.sync_params()
        seq = []
        it = 1
        for K in (1, 2, 13, 60, 225):
            norms, ms, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
            it += K
            seq.append(state(mod) + [norms.copy()])
        mod.state_save()
        best = 1e9
        for rep in range(7):
            mod.state_restore()
            norms, ms, _ = mod.run_fixed(200, first_major_iter=it, beta=1e3)
            best = min(best, ms / 200)
        seq.append(state(mod) + [norms.copy()])
        outs[res] = (seq, best)
        mod.close()
    same = [all(np.array_equal(x, y) for x, y in zip(a, b)) for a, b in zip(outs["0"][0], outs["1"][0])]
    print("%s: bitwise equal after K = 1, 2, 13, 60, 225, +200: %s | two launches per iteration %.4f ms/iteration (%.0f it/s), resident %.4f ms/iteration (%.0f it/s)"
          % (case, same, outs["0"][1], 1.0 / outs["0"][1] * 1e3, outs["1"][1], 1.0 / outs["1"][1] * 1e3), flush=True)
    if not all(same):
        for k, (a, b) in enumerate(zip(outs["0"][0], outs["1"][0])):
            print("   stage", k, [bool(np.array_equal(x, y)) for x, y in zip(a, b)], [float(np.max(np.abs(x.astype(float) - y.astype(float)))) for x, y in zip(a, b)])
