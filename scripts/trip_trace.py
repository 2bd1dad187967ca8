#!/usr/bin/env python3
"""GPU diagnostic: solver-loop trips of EVERY branch in EVERY inner iteration of a whole solve (replayed one iteration per call from the
states of admm_two_level's own loop, same trajectory bit for bit) -> gpurun_out/trips_<case>.npz for scripts/slack_model.py.
    python scripts/trip_trace.py [case] [outdir]"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import exaadmm_loader
import bench as B

pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
lib = pkg.library()
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
params = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0)


def model():
    env = pkg.AdmmEnv(case, params["rho_pq"], params["rho_va"], tight_factor=0.99, verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    env.params.scale = params["scale"]
    mod.sync_params()
    return env, mod


def norm_z(mod):
    v = C.c_double()
    pkg._check(lib.exaadmm_norm_z(mod.handle, C.byref(v)), "norm_z")
    return v.value


env_a, a = model()
env_b, b = model()
sqrt_d = math.sqrt(2 * g["ngen"] + 8 * g["nline"])
beta, theta, inc_c, outer_eps = 1e3, 0.8, 6.0, 2e-4
rows, marks, ms_rows = [], [], []
for outer in range(1, 21):
    nzp = norm_z(a)
    inner, res = a.run_inner(outer, 1000, beta)
    for done in range(inner):
        norms, ms, _ = b.run_fixed(1, first_major_iter=done + 1, beta=beta)
        _, _, t = b.branch_counters()
        rows.append(np.minimum(t, 255).astype(np.uint8))
        marks.append(outer)
        ms_rows.append(ms)
    assert np.array_equal(a.solution.u_curr, b.solution.u_curr), "replay left the solve's trajectory"
    if res[3] <= sqrt_d * outer_eps:
        break
    for m in (a, b):
        pkg._check(lib.exaadmm_update_lz(m.handle, beta), "update_lz")
    if res[2] > theta * nzp:
        beta = min(inc_c * beta, 1e24)
os.makedirs(out, exist_ok=True)
np.savez_compressed(os.path.join(out, "trips_%s.npz" % case), trips=np.stack(rows), outer=np.array(marks), ms=np.array(ms_rows),
                    brBusIdx=g["brBusIdx"], gen_bus=g["gen_bus"], nbus=g["nbus"])
print("%d iterations, %d branches, trips max %d" % (len(rows), g["nline"], np.stack(rows).max()))
