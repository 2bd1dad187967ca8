#!/usr/bin/env python3
"""GPU experiment: would packing the branches that need several solver-loop trips into few wavefronts pay?  The branch launch's cost is the sum
over wavefronts of the MAXIMUM trip count of their 64 lanes; a storage order that puts last iteration's multi-trip branches next to each other
would cut that sum -- if the set of such branches persists over iterations.  Measures the persistence and the potential.
usage: exp_resort_potential.py [case] [warm]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, exaadmm_loader
pkg = exaadmm_loader.load_package(); syn = exaadmm_loader.load_submodule("synthetic")
case = sys.argv[1] if len(sys.argv) > 1 else "case_ACTIVSg70k"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rho = {"case13659pegase": (5e1, 5e3), "case_ACTIVSg70k": (3e4, 3e5), "synthetic1M": (5e1, 5e3)}[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv("x", rho[0], rho[1], verbose=0, grid=g); mod = pkg.ModelAcopf(env)
env.params.scale = 1e-5 if case == "case_ACTIVSg70k" else 1e-4; mod.sync_params()
L = g["nline"]
perm = None
def ticks(n, first):
    mod.run_fixed(n, first_major_iter=first, beta=1e3)
    return mod.branch_counters()[2].astype(np.int64)
def wave_cost(t, order):
    tt = t[order]
    pad = (-len(tt)) % 64
    tt = np.concatenate([tt, np.zeros(pad, dtype=tt.dtype)]).reshape(-1, 64)
    return tt.max(axis=1).sum(), len(tt)
done = 0
for stage in (warm, 1, 10, 50, 200):
    t = ticks(stage, done + 1); done += stage
    # device storage order is bus-sorted; branch_counters() returns reference order -> approximate the storage order by the from-bus sort
    bb = np.asarray(g["brBusIdx"]).reshape(L, 2)
    store = np.argsort(bb[:, 0], kind="stable")
    if perm is None:
        heavy0 = t >= 2
        order_sorted = np.concatenate([store[heavy0[store]], store[~heavy0[store]]])
        perm = order_sorted
        print("after %d iterations: trips histogram %s" % (done, np.bincount(t)[:9]))
    c_nat, nw = wave_cost(t, store)
    c_srt, _ = wave_cost(t, perm)
    c_ideal, _ = wave_cost(t, np.argsort(-t, kind="stable"))
    heavy = t >= 2
    print("iteration %5d: multi-trip branches %6d (%.2f %%), still in the set of iteration %d: %.1f %% | sum over waves of max trips: storage order %d, "
          "sorted by iteration-%d counts %d, sorted by its own counts %d (waves %d)" % (done, heavy.sum(), 100.0 * heavy.mean(), warm, 100.0 * (heavy & heavy0).sum() / max(heavy.sum(), 1),
                                                                                         c_nat, warm, c_srt, c_ideal, nw))
