"""Experiment: does launching the heavy wavefronts of the branch kernel first (longest-processing-time order) shorten a
launch that needs more than one round of the 1024 SIMDs?  Uses the lane-order hook at WAVE granularity (whole 64-branch
waves are permuted, so loads stay coalesced).  Needs EXAADMM_NO_ZLFUSE=1 (the hook is not honoured by the fused variant)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exaadmm_loader
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
case = sys.argv[1] if len(sys.argv) > 1 else "case_ACTIVSg70k"
RHO = {"case_ACTIVSg70k": (3e4, 3e5), "synthetic1M": (5e1, 5e3), "case13659pegase": (5e1, 5e3)}
kw = {"cost": "market"} if case == "case_ACTIVSg70k" else {}
g = syn.sized_like(case, seed=0, **kw) if case != "synthetic1M" else syn.synthetic_grid(50000, 1000000, 500000, seed=0)
rho = RHO[case]
env = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
L = g["nline"]
mod.run_fixed(300, first_major_iter=1, beta=1e3)
def timed(n=100):
    _, ms, msb = mod.run_fixed(n, first_major_iter=301, beta=1e3, time_branch=True)
    return ms / n, (msb or 0) / n
base = [timed() for _ in range(3)]
a, m, t = mod.branch_counters()
nw = (L + 63) // 64
tp = np.zeros(nw * 64, dtype=np.int64); tp[:L] = t
wmax = tp.reshape(nw, 64).max(axis=1)
print("waves", nw, "ticks/branch mean %.2f max %d; wave max: mean %.2f, hist" % (t.mean(), t.max(), wmax.mean()), np.bincount(wmax)[:16].tolist())
def order_from(wave_perm):
    o = (wave_perm[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
    return o[o < L] if False else o
# the hook wants a permutation of the slots 0..La-1: keep the (partial) last wave last
full = nw - (1 if L % 64 else 0)
perm = np.argsort(-wmax[:full], kind="stable")
o = np.concatenate([(perm[:, None] * 64 + np.arange(64)[None, :]).reshape(-1), np.arange(full * 64, L)]).astype(np.int32)
assert np.array_equal(np.sort(o), np.arange(L))
mod.set_branch_lane_order(o)
lpt = [timed() for _ in range(3)]
rev = np.argsort(wmax[:full], kind="stable")
o2 = np.concatenate([(rev[:, None] * 64 + np.arange(64)[None, :]).reshape(-1), np.arange(full * 64, L)]).astype(np.int32)
mod.set_branch_lane_order(o2)
spt = [timed() for _ in range(3)]
mod.set_branch_lane_order(None)
again = [timed() for _ in range(2)]
print(json.dumps({"case": case, "waves": int(nw), "ms_per_iter[total,branch]": {"natural": base, "heavy_first": lpt, "light_first": spt, "natural_again": again}}))
