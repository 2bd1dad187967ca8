#!/usr/bin/env python3
"""Experiment: does grouping branches by their previous solver work cut k_branch time?
A/B from the same saved ADMM state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
start = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rho = {"case1354pegase": (1e1, 1e3), "case13659pegase": (5e1, 5e3), "case_ACTIVSg70k": (3e4, 3e5), "synthetic1M": (5e1, 5e3)}[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
mod.run_fixed(start, first_major_iter=1, beta=1e3)
names = ("u_curr", "v_curr", "z_curr", "l_curr", "lz", "rho", "z_prev")
state = {n: getattr(mod.solution, n) for n in names}
mb = mod.membuf.copy()
def restore():
    for n in names:
        setattr(mod.solution, n, state[n])
    mod.membuf = mb
def run(order, n=20):
    restore()
    mod.set_branch_lane_order(order)
    norms, ms, msb = mod.run_fixed(n, first_major_iter=start + 1, beta=1e3, time_branch=True)
    return msb, ms / n, norms[0]
r = run(None)
print("identity        : k_branch %.4f ms  iter %.4f ms primres %.12f" % r)
restore(); mod.set_branch_lane_order(None); mod.run_fixed(1, first_major_iter=start + 1, beta=1e3)
a, m, t = mod.branch_counters()
print("ticks hist", np.bincount(np.minimum(t, 12)), "auglag hist", np.bincount(np.minimum(a, 8)))
# device position of a branch (bus-sorted storage): keep it as the secondary key so that lanes of one
# bucket still read (mostly) consecutive plane elements
fr = np.asarray(env.data["brBusIdx"]).reshape(-1, 2)[:, 0]
pos = np.empty(len(fr), dtype=np.int64); pos[np.argsort(fr, kind="stable")] = np.arange(len(fr))
wmax = lambda order: np.maximum.reduceat(t[order], np.arange(0, len(t), 64)).sum()
print("sum over waves of max ticks: identity(device order) %d, sorted by ticks %d, mean ticks x waves %.0f" % (
    wmax(np.argsort(pos)), wmax(np.argsort(t.astype(np.int64) * (1 << 32) + pos)), t.mean() * len(t) / 64))
for key_name, key in (("ticks,pos", t.astype(np.int64) * (1 << 32) + pos), ("-ticks,pos", (1000 - t.astype(np.int64)) * (1 << 32) + pos),
                      ("-min(ticks,8),pos", (1000 - np.minimum(t.astype(np.int64), 8)) * (1 << 32) + pos),                       ):
    order = np.argsort(key, kind="stable")
    r = run(order)
    print("sorted by %-12s: k_branch %.4f ms  iter %.4f ms primres %.12f" % ((key_name,) + r))
r = run(None)
print("identity again  : k_branch %.4f ms  iter %.4f ms primres %.12f" % r)
