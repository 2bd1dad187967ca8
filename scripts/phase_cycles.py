#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DEXA_PROFILE_PHASES, EXAADMM_HIP_LIB=...): where do the
cycles of a branch-solver tick go?  Prints per-phase shader-clock cycles for light and heavy branches."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
L = pkg.library()
L.exaadmm_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
case = sys.argv[2] if len(sys.argv) > 2 else "case13659pegase"   # usage: phase_cycles.py [warm iterations] [case]
rho, scale = {"case_ACTIVSg70k": ((3e4, 3e5), 1e-5)}.get(case, ((5e1, 5e3), 1e-4))
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
env.params.scale = scale
mod.sync_params()
nwarm = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mod.run_fixed(nwarm, first_major_iter=1, beta=1e3)
for rep in range(2):
    norms, ms, msb = mod.run_fixed(1, first_major_iter=nwarm + 1 + rep, beta=1e3, time_branch=True)
    a, m, t = mod.branch_counters()
    Lp = ((g["nline"] + 63) // 64) * 64
    buf = np.zeros(13 * Lp, dtype=np.uint64)
    pkg._check(L.exaadmm_debug_phase_cycles(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "phase cycles")
    cyc = buf.reshape(13, Lp)[:, :g["nline"]].astype(np.float64)
    names = ["fgh+newx", "model step", "f(trial)", "ratio test", "auglag step", "icfs", "trpcg", "prsrch", "#faces", "#cg", "cauchy", "spcg", "#lone ticks"]
    print("k_branch %.3f ms; ticks: median %d max %d; auglag max %d" % (msb, np.median(t), t.max(), a.max()))
    light = cyc[12] == 2
    heavy = cyc[12] >= 4
    one = cyc[12] == 1
    for nm, sel in (("1 lone tick", one), ("2 lone ticks", light), (">=4 lone ticks", heavy)):
        if not sel.any():
            continue
        tot = cyc[:5, sel].sum(axis=0).mean()
        print("  %-22s n=%5d ticks %.1f total cyc %.0f (%.1f us @2.4GHz)  per tick %.0f | " % (nm, sel.sum(), t[sel].mean(), tot, tot / 2400.0, tot / t[sel].mean())
              + ", ".join("%s %.0f" % (names[q], cyc[q, sel].mean()) for q in range(13)))
