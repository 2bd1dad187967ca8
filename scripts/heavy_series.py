#!/usr/bin/env python3
"""GPU diagnostic (library built with -DEXA_PROFILE_PHASES -DEXA_HEAVY_SERIES, EXAADMM_PROF_RESIDENT=1): the LAST 16 iterations of one resident
launch, wavefront by wavefront -- heavy wavefronts (top, top wait, solve, decision + stores, arrivals, trips) and ordinary tiles (top, wait for
decision and buses, z / l pass + contribution, solve, stores + arrivals, largest trip count).  Who sets the pace of an iteration?
    python scripts/heavy_series.py [case] [warm iterations] [launch length]"""
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
K = int(sys.argv[3]) if len(sys.argv) > 3 else 48
par = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv(case, par["rho_pq"], par["rho_va"], tight_factor=0.99, verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
env.params.scale = par["scale"]
mod.sync_params()
mod.run_fixed(warm, first_major_iter=1, beta=1e3)
Lp = ((g["nline"] + 63) // 64) * 64
buf = np.zeros(2 * Lp, dtype=np.uint64)
L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
norms, ms, _ = mod.run_fixed(K, first_major_iter=warm + 1, beta=1e3)
pkg._check(L.exaadmm_debug_wave_timeline(mod.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size), "timeline")
form = mod.resident_form()
print("# %s: %.2f us per iteration (device time / %d), resident form %s" % (case, 1e3 * ms / K, K, form))
tl = buf.reshape(-1, 128)
W = 16
m20 = np.uint64((1 << 20) - 1)
h = tl[:, 32:32 + 3 * W].reshape(tl.shape[0], W, 3)
t = tl[:, 80:80 + 3 * W].reshape(tl.shape[0], W, 3)
hon = h[:, 0, 0] != 0
hon[:16] = False   # (the bus launch of the diagnostic build stamps into the first 2 048 entries: blocks 0 .. 15)
ton = t[:, 0, 0] != 0
ton[:16] = False
h, t = h[hon], t[ton]
htop = h[:, :, 0].astype(np.float64)
hsolve = (h[:, :, 1] >> np.uint64(8)).astype(np.int64)
htrips = (h[:, :, 1] & np.uint64(255)).astype(np.int64)
htopw = (h[:, :, 2] & m20).astype(np.int64)
hdec = ((h[:, :, 2] >> np.uint64(20)) & m20).astype(np.int64)
harr = ((h[:, :, 2] >> np.uint64(40)) & m20).astype(np.int64)
hkept = (h[:, :, 2] >> np.uint64(60)).astype(np.int64)
ttop = t[:, :, 0].astype(np.float64)
twait = (t[:, :, 1] & m20).astype(np.int64)
tzl = ((t[:, :, 1] >> np.uint64(20)) & m20).astype(np.int64)
tsolve = ((t[:, :, 1] >> np.uint64(40)) & m20).astype(np.int64)
tarr = (t[:, :, 2] & m20).astype(np.int64)
ttrips = ((t[:, :, 2] >> np.uint64(20)) & np.uint64(4095)).astype(np.int64)
print("# %d heavy wavefronts, %d tiles with stamps; iterations %d .. %d of the launch" % (h.shape[0], t.shape[0], K - W + 1, K))
hend = htop + htopw + 4500 + hsolve + hdec + harr        # (~ when the heavy wavefront's arrivals were done; 4 500: its inputs)
tend = ttop + twait + tzl + tsolve + tarr
for i in range(W - 1):
    wh, wt = int(np.argmax(hend[:, i])), int(np.argmax(tend[:, i]))
    t0 = min(htop[:, i].min(), ttop[:, i].min())
    print("  %2d: period (median of tops) %6.0f | last heavy: w %3d end %+7.0f (top %+6.0f wait %5d solve %6d x%d dec+stores %5d arr %5d kept %d) | last tile: w %3d end %+7.0f (top %+6.0f wait %5d zl %5d solve %6d x%d arr %5d)"
          % (K - W + 1 + i, np.median(np.concatenate([htop[:, i + 1] - htop[:, i], ttop[:, i + 1] - ttop[:, i]])), wh, hend[wh, i] - t0, htop[wh, i] - t0, htopw[wh, i], hsolve[wh, i], htrips[wh, i], hdec[wh, i],
             harr[wh, i], hkept[wh, i], wt, tend[wt, i] - t0, ttop[wt, i] - t0, twait[wt, i], tzl[wt, i], tsolve[wt, i], ttrips[wt, i], tarr[wt, i]))
print("# per iteration, the heavy wavefront with the LONGEST solve: solve clocks x trips | its top wait (kept bits) | decision + stores | arrivals | its period")
for i in range(W - 1):
    w = int(np.argmax(hsolve[:, i]))
    print("  %2d: w %3d solve %6d x%d | top wait %6d (kept %d) | dec+stores %6d | arr %5d | period %7.0f" % (K - W + 1 + i, w, hsolve[w, i], htrips[w, i], htopw[w, i], hkept[w, i], hdec[w, i], harr[w, i], htop[w, i + 1] - htop[w, i]))
print("# tiles: largest trip count per iteration: %s" % ttrips.max(axis=0).tolist())
print("# tiles: median over tiles of wait / zl / solve / arr: %.0f / %.0f / %.0f / %.0f clocks" % (np.median(twait), np.median(tzl), np.median(tsolve), np.median(tarr)))
print("# heavy: median over wavefronts of top wait / solve / decision + stores / arrivals: %.0f / %.0f / %.0f / %.0f clocks" % (np.median(htopw), np.median(hsolve), np.median(hdec), np.median(harr)))
