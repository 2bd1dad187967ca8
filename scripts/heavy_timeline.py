#!/usr/bin/env python3
"""GPU diagnostic (library built with -DEXA_PROFILE_PHASES, EXAADMM_HIP_LIB=.../libexaadmm_hip_prof.so, EXAADMM_PROF_RESIDENT=1):
what does a HEAVY wavefront of the resident loop (64 clones of one branch: the steady state's critical path) do between two solves?
Every heavy wavefront stamps (shader clock) its last two iterations: top, decision + flags seen, inputs formed, solve over,
stores acknowledged, arrivals done.
    python scripts/heavy_timeline.py [case] [warm iterations]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import exaadmm_loader
import bench as B

pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
L = pkg.library()
L.exaadmm_debug_wave_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
par = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv(case, par["rho_pq"], par["rho_va"], tight_factor=0.99, verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
env.params.scale = par["scale"]
mod.sync_params()
assert mod.loop_kind() == 2, "needs the resident loop (EXAADMM_PROF_RESIDENT=1 with the diagnostic build)"
mod.run_fixed(warm, first_major_iter=1, beta=1e3)
Lp = ((g["nline"] + 63) // 64) * 64
buf = np.zeros(2 * Lp, dtype=np.uint64)
L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)   # (clears)
K = 40
norms, ms, _ = mod.run_fixed(K, first_major_iter=warm + 1, beta=1e3)
pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "timeline")
print("# %s: %.2f us per iteration (device time / %d), resident form %s" % (case, 1e3 * ms / K, K, mod.resident_form() if hasattr(mod, "resident_form") else "?"))
tl = buf.reshape(-1, 128)
rows = []
for t in range(tl.shape[0]):
    for grp in range(4):
        a, b = tl[t, 32 + 16 * grp:40 + 16 * grp].astype(np.float64), tl[t, 40 + 16 * grp:48 + 16 * grp].astype(np.float64)
        if a[7] == 0 or b[7] == 0:
            continue
        last, prev = (a, b) if a[7] > b[7] else (b, a)
        x = tl[t, 96:104].astype(np.float64) if a[7] > b[7] else tl[t, 104:112].astype(np.float64)
        rows.append((last, prev, x if grp == 0 else None))
clk = 2.4e3
names = ["top -> decision + flags seen", "-> bus results in, v, z/l, start point", "SOLVE", "-> results / hand-over / terms stored (acked)", "-> arrivals + bus updates done",
         "-> top of the next iteration"]
print("# %d heavy wavefronts with stamps" % len(rows))
if rows:
    per = np.array([l[0] - p[0] for l, p, _ in rows])
    trips = np.array([int(l[6]) >> 8 for l, p, _ in rows])
    kept = np.array([int(l[6]) & 3 for l, p, _ in rows])
    seg = np.array([[l[1] - l[0], l[2] - l[1], l[3] - l[2], l[4] - l[3], l[5] - l[4], l[0] - p[5]] for l, p, _ in rows])
    print("# trips of the last solve: " + ", ".join("%d trips x %d" % (t, int((trips == t).sum())) for t in sorted(set(trips.tolist()))))
    print("# own bus updates kept at the top of the last iteration (bit 0 from, bit 1 to): " + ", ".join("%d x %d" % (k, int((kept == k).sum())) for k in sorted(set(kept.tolist()))))
    for sel, label in ((trips == trips.max(), "heavy wavefronts with the most trips (%d)" % trips.max()), (trips >= 0, "all heavy wavefronts")):
        print("## %s: n = %d, iteration period median %.0f clocks = %.1f us" % (label, int(sel.sum()), np.median(per[sel]), np.median(per[sel]) / clk))
        for k, n in enumerate(names):
            v = seg[sel, k]
            print("  %-50s median %8.0f clocks %6.2f us   (min %.0f, max %.0f)" % (n, np.median(v), np.median(v) / clk, v.min(), v.max()))
    w = int(np.argmax(seg[:, 2]))
    print("## the heavy wavefront with the longest solve: %d trips, period %.0f clocks" % (trips[w], per[w]))
    for k, n in enumerate(names):
        print("  %-50s %8.0f clocks %6.2f us" % (n, seg[w, k], seg[w, k] / clk))
    sub = np.array([[x[0] - l[3], l[4] - x[0], x[1] - l[4], x[2] - x[1], x[3] - x[2]] for l, p, x in rows if x is not None and x[1] > 0 and (int(l[6]) & 1)])
    if len(sub):
        print("## sub-stamps (heavy wavefronts whose lane 0 completed its from-bus, n = %d), median clocks:" % len(sub))
        for k, n in enumerate(["solve over -> hand-over / planes / diagnostics acknowledged", "-> terms acknowledged", "-> arrival counter back", "-> terms loaded, update computed", "-> result stores acknowledged"]):
            print("  %-62s %8.0f clocks %6.2f us" % (n, np.median(sub[:, k]), np.median(sub[:, k]) / clk))
