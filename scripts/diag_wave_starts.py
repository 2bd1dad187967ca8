#!/usr/bin/env python3
"""Diagnostic (library built with -DEXA_PROFILE_PHASES, EXAADMM_HIP_LIB=...): WHEN do the wavefronts of the branch launch start and end?
s_memtime stamps per wavefront (first instruction, stores landed), relative to the earliest start.  usage: diag_wave_starts.py [case] [warm]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package(); syn = exaadmm_loader.load_submodule("synthetic")
L = pkg.library()
L.exaadmm_debug_wave_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
nwarm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = syn.sized_like(case, seed=0)
rho = {"case_ACTIVSg70k": (3e4, 3e5)}.get(case, (5e1, 5e3))
env = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g); mod = pkg.ModelAcopf(env)
env.params.scale = 1e-5 if case == "case_ACTIVSg70k" else 1e-4; mod.sync_params()
mod.run_fixed(nwarm, first_major_iter=1, beta=1e3)
Lp = ((g["nline"] + 63) // 64) * 64
nw = Lp // 64
buf = np.zeros(2 * Lp, dtype=np.uint64)
pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "tl")
for rep in range(2):
    norms, ms, msb = mod.run_fixed(1, first_major_iter=nwarm + 1 + rep, beta=1e3, time_branch=True)
    pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "tl")
    tl = (buf.reshape(nw, 128) & np.uint64(0x00ffffffffffffff)).astype(np.int64)
    ok = np.arange(nw) >= 32                      # (the bus launch's own stamps overwrite the first wavefronts' rows)
    start, stored = tl[ok, 122], tl[ok, 126]
    good = (start > 0) & (stored > start)
    start, stored = start[good], stored[good]
    t0 = start.min()
    q = [0, 10, 25, 50, 75, 90, 99, 100]
    print("%s, k_branch %.1f us (HIP events), %d wavefronts with stamps" % (case, 1e3 * msb, good.sum()))
    print("  wave START after the first one, percentiles %s: %s clocks" % (q, np.percentile(start - t0, q).astype(int).tolist()))
    print("  wave END   after the first start,               %s clocks" % np.percentile(stored - t0, q).astype(int).tolist())
    print("  wave duration                                   %s clocks" % np.percentile(stored - start, q).astype(int).tolist())
    hw = tl[ok][good][:, 121]
    # gfx9 HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0] of the second register
    slot = ((hw >> 32) & 0xf) * 100000 + ((hw >> 13) & 7) * 10000 + ((hw >> 12) & 1) * 1000 + ((hw >> 8) & 0xf) * 10 + ((hw >> 4) & 3)
    gaps, per_slot = [], []
    for sl in np.unique(slot):
        m = slot == sl
        o = np.argsort(start[m])
        st_, en_ = start[m][o], stored[m][o]
        per_slot.append(m.sum())
        gaps.extend((st_[1:] - en_[:-1]).tolist())
    gaps = np.asarray(gaps)
    print("  SIMD slots seen %d, wavefronts per slot min / median / max %d / %d / %d" % (len(per_slot), min(per_slot), np.median(per_slot), max(per_slot)))
    if len(gaps):
        print("  gap on a SIMD between a wavefront's stores landing and its successor's first instruction, percentiles %s: %s clocks (negative: overlap)" % (q, np.percentile(gaps, q).astype(int).tolist()))
        print("  sum of wavefront durations %.0f M clocks, sum of gaps %.0f M clocks" % ((stored - start).sum() / 1e6, gaps.sum() / 1e6))
    ticks = (tl[ok][good][:, :120] > 0).sum(axis=1) - 1
    print("  trips per wavefront: %s" % np.bincount(np.maximum(ticks, 0))[:10].tolist())
