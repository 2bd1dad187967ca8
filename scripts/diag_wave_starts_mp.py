#!/usr/bin/env python3
"""Diagnostic build (EXAADMM_HIP_LIB=exaadmm.jl_amd/libexaadmm_hip_prof.so): the branch launch of the multi-period model on the headline grid,
T periods -- when do its wavefronts start and end, which one ends last, how many solver trips did it run?  usage: diag_wave_starts_mp.py T [T ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
import bench as B
pkg = exaadmm_loader.load_package(); syn = exaadmm_loader.load_submodule("synthetic")
L = pkg.library()
L.exaadmm_debug_wave_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
case = "case13659pegase"
params = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0)
for T in [int(a) for a in sys.argv[1:]] or [3, 4]:
    env = pkg.AdmmEnv(case, params["rho_pq"], params["rho_va"], tight_factor=0.99, verbose=0, grid=g)
    lvl = 1.0 + 0.01 * np.sin(np.arange(T) * 2 * np.pi / max(T, 2))
    load = (np.stack([env.data["Pd"] * a for a in lvl], 1), np.stack([env.data["Qd"] * a for a in lvl], 1))
    mod = pkg.ModelMpacopf(env, load, end_period=T)
    env.params.scale = params["scale"]; mod.sync_params()
    mod.run_fixed(320, first_major_iter=1, beta=1e3)
    Lx = T * int(g["nline"])
    Lp = ((Lx + 63) // 64) * 64
    nw = Lp // 64
    buf = np.zeros(2 * Lp, dtype=np.uint64)
    pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "tl")
    for rep in range(2):
        norms, ms, msb = mod.run_fixed(1, first_major_iter=321 + rep, beta=1e3, time_branch=True)
        pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "tl")
        tl = (buf.reshape(nw, 128) & np.uint64(0x00ffffffffffffff)).astype(np.int64)
        start, stored = tl[:, 122], tl[:, 126]
        good = (start > 0) & (stored > start)
        idx = np.flatnonzero(good)
        start, stored = start[good], stored[good]
        xcc = (tl[good][:, 121] >> 32) & 0xf       # (every XCD counts its own clock: stamps are taken relative to the XCD's first wavefront)
        t0 = np.zeros_like(start)
        for x in np.unique(xcc):
            t0[xcc == x] = start[xcc == x].min()
        ticks = (tl[good][:, :120] > 0).sum(axis=1) - 1
        last = np.argsort(stored - t0)[-5:][::-1]
        q = [0, 50, 90, 99, 100]
        print("T = %d: k_branch %.1f us by HIP events; %d of %d branch wavefronts stamped (+ %d generator wavefronts without stamps; 1 024 SIMD slots)" %
              (T, 1e3 * msb, good.sum(), nw, (T * int(g["ngen"]) + 63) // 64))
        print("   wave start after the first start, percentiles %s: %s clocks; started later than 10 k clocks: %d" % (q, np.percentile(start - t0, q).astype(int).tolist(), int(((start - t0) > 10000).sum())))
        print("   wave end, percentiles %s: %s clocks; launch by events = %d clocks at 2.4 GHz" % (q, np.percentile(stored - t0, q).astype(int).tolist(), int(msb * 2.4e6)))
        print("   the five wavefronts that end last: " + "; ".join("block %d start %d end %d trips %d" % (idx[i], (start - t0)[i], (stored - t0)[i], ticks[i]) for i in last))
    mod.close()
