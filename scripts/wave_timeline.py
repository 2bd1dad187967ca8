#!/usr/bin/env python3
"""Diagnostic (library built with -DEXA_PROFILE_PHASES, EXAADMM_HIP_LIB=...): per-wavefront tick
timeline of k_branch -- how many lanes are still active at each tick and what each tick costs."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
L = pkg.library()
L.exaadmm_debug_wave_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
g = syn.sized_like(case, seed=0)
rho = {"case_ACTIVSg70k": (3e4, 3e5)}.get(case, (5e1, 5e3))
env = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
env.params.scale = 1e-5 if case == "case_ACTIVSg70k" else 1e-4
mod.sync_params()
nwarm = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mod.run_fixed(nwarm, first_major_iter=1, beta=1e3)
Lp = ((g["nline"] + 63) // 64) * 64
nw = Lp // 64
buf = np.zeros(2 * Lp, dtype=np.uint64)
pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "tl")
for rep in range(3):
    norms, ms, msb = mod.run_fixed(1, first_major_iter=nwarm + 1 + rep, beta=1e3, time_branch=True)
    pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "tl")
    tl = buf.reshape(nw, 128).copy()
    tl[:16] = 0   # (the diagnostic build's bus kernel stamps its first 256 blocks into the first 2 048 entries: wavefronts 0-15 are lost)
    clk = (tl & np.uint64(0x00ffffffffffffff)).astype(np.int64)
    act = (tl >> np.uint64(56)).astype(np.int64)
    stamps = clk[:, 123:127].copy()
    t_start = clk[:, 122].copy()
    clk = clk.copy(); clk[:, 122:127] = 0; act = act.copy(); act[:, 122:127] = 0
    nt = np.maximum((clk > 0).sum(axis=1) - 1, 0)          # ticks of the wave
    dur = np.array([clk[w, nt[w]] - clk[w, 0] for w in range(nw)])
    t0 = clk[:, 0][clk[:, 0] > 0].min()
    end = np.array([clk[w, nt[w]] for w in range(nw)]).max()
    ok = (stamps > 0).all(axis=1) & (t_start > 0)
    if ok.any():
        print("  launch, from the stamps: first wave start -> last store landed %.0f clocks = %.1f us at 2.4 GHz (kernel by HIP events %.1f us); slowest wave start -> stores landed %.0f"
              % (stamps[ok, 3].max() - t_start[ok].min(), (stamps[ok, 3].max() - t_start[ok].min()) / 2400.0, msb * 1e3, (stamps[ok, 3] - t_start[ok]).max()))
        pro_load = stamps[ok, 1] - stamps[ok, 0]
        first_tick = clk[ok, 0] - stamps[ok, 1]
        epi = stamps[ok, 3] - stamps[ok, 2]
        total = stamps[ok, 3] - stamps[ok, 0]
        pre = stamps[ok, 0] - t_start[ok]
        print("  wave start (first instruction) -> entry (loop control + launch geometry read): median %.0f max %.0f clocks; first wave start -> last wave start %.0f clocks"
              % (np.median(pre), pre.max(), t_start[ok].max() - t_start[ok].min()))
        print("  in-kernel stamps per wave (clocks): entry->loads landed median %.0f max %.0f | loads landed->first tick median %.0f | solve end->stores landed median %.0f max %.0f | entry->stores landed median %.0f max %.0f"
              % (np.median(pro_load), pro_load.max(), np.median(first_tick), np.median(epi), epi.max(), np.median(total), total.max()))
    span = float(end - t0)
    print("k_branch %.3f ms | waves %d | wave ticks median %d max %d | wave duration median %.0f mean %.0f max %.0f clocks | first start..last end %.0f clocks "
          "(%.2f GHz if that is the kernel time) | sum of wave durations / (1024 SIMDs x span) = %.2f"
          % (msb, nw, np.median(nt), nt.max(), np.median(dur), dur.mean(), dur.max(), span, span / (msb * 1e6), dur.sum() / (1024.0 * span)))
    # per tick index: mean active lanes and mean cost, over all waves that reach it
    print("  tick: waves, mean active lanes, mean cycles")
    for k in list(range(0, 12)) + [15, 20, 25, 30, 40]:
        sel = nt > k
        if sel.sum() == 0:
            break
        c = clk[sel, k + 1] - clk[sel, k]
        print("   %3d: %4d  %5.1f  %7.0f" % (k + 1, sel.sum(), act[sel, k].mean(), c.mean()))
    # cost vs active-lane count
    rows = []
    for w in range(nw):
        for k in range(nt[w]):
            rows.append((act[w, k], clk[w, k + 1] - clk[w, k]))
    rows = np.array(rows)
    print("  active lanes -> mean tick cycles (count)")
    for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 63), (64, 64)):
        m = (rows[:, 0] >= lo) & (rows[:, 0] <= hi)
        if m.sum():
            print("   %2d-%2d: %7.0f (%d)" % (lo, hi, rows[m, 1].mean(), m.sum()))
    # the slowest waves
    for w in np.argsort(-dur)[:3]:
        k = nt[w]
        print("  slow wave %d: %d ticks, %d cycles; active lanes per tick: %s" % (w, k, dur[w], " ".join(str(a) for a in act[w, :k])))
        print("      cycles per tick: %s" % " ".join(str(int(c)) for c in (clk[w, 1:k + 1] - clk[w, :k])))
