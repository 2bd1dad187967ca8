#!/usr/bin/env python3
"""GPU: multi-period model on the headline grid, T = 2, 3, 4, 6 -- per T the branch launch's duration next to WHAT IT RAN: wavefronts of the
launch against the chip's 1 024 SIMD slots, and the solver trips of its branches (largest count, per-period largest, how many branches need
>= 4).  VERDICT r04 item 6: why is T = 3 (110 us) slower than T = 4 (98 us)?   usage: mp_trip_census.py [T ...]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
import bench as B
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
case = "case13659pegase"
params = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0)
Ts = [int(a) for a in sys.argv[1:]] or [2, 3, 4, 6]
for T in Ts:
    env = pkg.AdmmEnv(case, params["rho_pq"], params["rho_va"], tight_factor=0.99, verbose=0, grid=g)
    lvl = 1.0 + 0.01 * np.sin(np.arange(T) * 2 * np.pi / max(T, 2))
    load = (np.stack([env.data["Pd"] * a for a in lvl], 1), np.stack([env.data["Qd"] * a for a in lvl], 1))
    mod = pkg.ModelMpacopf(env, load, end_period=T)
    env.params.scale = params["scale"]; mod.sync_params()
    beta = env.params.initial_beta
    mod.run_fixed(320, first_major_iter=1, beta=beta)
    mod.state_save()
    ms = []
    for rep in range(5):
        mod.state_restore()
        _, tot, msb = mod.run_fixed(20, first_major_iter=321, beta=beta, time_branch=True)
        ms.append((tot / 20.0, msb))
    L1 = int(g["nline"]); Lx = T * L1
    a, m, t = (np.zeros(Lx, dtype=np.int32) for _ in range(3))
    ip = C.POINTER(C.c_int32)
    pkg._check(pkg.library().exaadmm_download_branch_counters(mod.handle, a.ctypes.data_as(ip), m.ctypes.data_as(ip), t.ctypes.data_as(ip), Lx), "counters")
    per = t.reshape(T, L1)
    nb, ng = (Lx + 63) // 64, (T * int(g["ngen"]) + 63) // 64
    print(json.dumps({"T": T, "load_levels": [round(float(x), 4) for x in lvl], "ms_per_iteration": round(float(np.median([x[0] for x in ms])), 4),
                      "k_branch_ms": round(float(np.median([x[1] for x in ms])), 4), "branch_wavefronts": nb, "generator_wavefronts": ng,
                      "wavefronts_per_simd_slot": round((nb + ng) / 1024.0, 2), "max_trips": int(t.max()), "max_trips_per_period": [int(x) for x in per.max(axis=1)],
                      "branches_with_4_or_more_trips_per_period": [int((x >= 4).sum()) for x in per], "max_auglag": int(a.max()), "loop_kind": mod.loop_kind()}))
    mod.close()
