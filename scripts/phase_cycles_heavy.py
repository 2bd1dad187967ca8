#!/usr/bin/env python3
"""Diagnostic (library built with -DEXA_PROFILE_PHASES: EXAADMM_HIP_LIB=exaadmm.jl_amd/libexaadmm_hip_prof.so EXAADMM_PROF_RESIDENT=1):
per-phase shader clocks of a solver trip ON A HEAVY WAVEFRONT of the resident loop (64 clones of one branch's solve, the Cauchy search's
candidates side by side) next to the trips an ordinary lane runs alone.  VERDICT r04 item 2 asked for these before anything else is
built on the heavy wavefronts.   usage: phase_cycles_heavy.py [case] [prep]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
import bench as B
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
L = pkg.library()
L.exaadmm_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
prep = int(sys.argv[2]) if len(sys.argv) > 2 else 320
params = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv(case, params["rho_pq"], params["rho_va"], tight_factor=0.99, verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
env.params.scale = params["scale"]; mod.sync_params()
mod.set_option("trust_resident", 1)
assert mod.loop_kind() == 2, "needs the resident loop (EXAADMM_PROF_RESIDENT=1 with the diagnostic build)"
mod.run_fixed(prep, first_major_iter=1, beta=1e3)
names = ["g,H (+f at a restart)", "model step", "f(trial)", "ratio test + g + H", "auglag step", "icfs", "trpcg", "prsrch", "#faces", "#cg", "cauchy", "spcg", "#trips counted"]
for rep in range(3):
    mod.run_fixed(20, first_major_iter=prep + 1 + 20 * rep, beta=1e3)
    form = mod.resident_form()
    a, m, t = mod.branch_counters()
    Lp = ((g["nline"] + 63) // 64) * 64
    buf = np.zeros(13 * Lp, dtype=np.uint64)
    pkg._check(L.exaadmm_debug_phase_cycles(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "phase cycles")
    raw = buf.reshape(13, Lp)
    # st_cyc is in STORAGE order; only the shape of the distribution is used here
    cyc = raw.astype(np.float64)
    heavy = cyc[12] >= 1000
    trips_h = cyc[12, heavy] - 1000
    print("run %d: form %s; branches on heavy wavefronts with a solve recorded: %d (trips: median %d, max %d)" % (rep, form, heavy.sum(), np.median(trips_h) if heavy.any() else 0, trips_h.max() if heavy.any() else 0))
    for nm, sel, ntr in (("heavy wavefront, >= 4 trips", heavy & (cyc[12] >= 1004), None), ("ordinary lane, >= 3 lone trips", (~heavy) & (cyc[12] >= 3), None)):
        if not sel.any():
            continue
        n = np.where(heavy[sel], cyc[12, sel] - 1000, cyc[12, sel])
        tot = cyc[:5, sel].sum(axis=0)
        print("  %-30s n=%4d  clocks per trip %.0f | per trip: " % (nm, sel.sum(), (tot / n).mean())
              + ", ".join("%s %.0f" % (names[q], (cyc[q, sel] / n).mean()) for q in (0, 1, 2, 3, 4, 10, 11, 5, 6, 7)))
