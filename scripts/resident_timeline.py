#!/usr/bin/env python3
"""GPU diagnostic (library built with -DEXA_PROFILE_PHASES, EXAADMM_HIP_LIB=.../libexaadmm_hip_prof.so, EXAADMM_PROF_RESIDENT=1):
what does the resident loop's SLOWEST wavefront do between two solves?  Every branch wavefront stamps (shader clock) in its
last two iterations: top of the iteration, loads landed, both buses ready, v + z / l pass + residual contribution done, solve over,
results and terms stored (acknowledged), arrivals (incl. the bus updates its lanes completed) done.
    python scripts/resident_timeline.py [case] [warm iterations]"""
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
nb = (g["nline"] + 63) // 64
tl = buf.reshape(-1, 128)[:nb].astype(np.float64)
a, b = tl[:, 0:8], tl[:, 8:16]                     # iteration parity 0 / 1 slots
last = np.where(a[:, 7:8] > b[:, 7:8], a, b)       # the later of the two iterations
prev = np.where(a[:, 7:8] > b[:, 7:8], b, a)
names = ["top -> loads landed", "loads landed -> both buses ready", "buses ready -> v, z/l pass, contribution", "SOLVE", "solve over -> results + terms stored (acked)",
         "stored -> arrivals + bus updates done", "arrivals done -> top of the next iteration"]
clk = 2.4e3   # clocks per us (shader clock ~2.4 GHz)
solve = last[:, 4] - last[:, 3]
w = int(np.argmax(solve))
period = last[:, 0] - prev[:, 0]
print("# %s, resident loop, iteration %d of a %d-iteration launch after %d warm iterations: %.2f us per iteration (device time / %d)" % (case, int(last[w, 7]) + 1, K, warm, 1e3 * ms / K, K))
print("# slowest wavefront (block %d): iteration period %.0f clocks = %.1f us" % (w, period[w], period[w] / clk))
seg = [last[w, k + 1] - last[w, k] for k in range(6)] + [last[w, 0] - prev[w, 6]]
for n, s in zip(names, seg):
    print("  %-58s %8.0f clocks  %6.2f us" % (n, s, s / clk))
print("  (sum %.0f clocks)" % sum(seg))
med = [np.median(last[:, k + 1] - last[:, k]) for k in range(6)]
print("# median wavefront: " + ", ".join("%s %.0f" % (n.split(" ->")[0] if n != "SOLVE" else n, s) for n, s in zip(names, med)) + " clocks; median solve %.1f us, median wait for its buses %.1f us" % (med[3] / clk, med[1] / clk))
