#!/usr/bin/env python3
"""GPU diagnostic: what does an inner iteration cost at every point of a WHOLE solve (the reference's metric: cumul /
time_overall, print_statistics.jl:12)?  A first model runs admm_two_level's own loop (exaadmm_run_inner per outer iteration) to learn how
many inner iterations every outer iteration takes; a second model replays exactly those iterations in fixed-count pieces of
10 (exaadmm_run_fixed from the same states: same trajectory bit for bit) and reports, per piece: device ms per iteration, the
largest solver-trip count of a branch and the number of lanes with >= 3 trips in the piece's last iteration.
    python scripts/iter_profile_solve.py [case] [piece]"""
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
piece = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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
print("# %s (%d branches), loop kind %d: device ms per iteration over the whole solve, pieces of %d iterations" % (case, g["nline"], b.loop_kind(), piece))
cumul, total_ms = 0, 0.0
for outer in range(1, 21):
    nzp = norm_z(a)
    assert norm_z(b) == nzp
    inner, res = a.run_inner(outer, 1000, beta)
    print("outer %d: %d inner iterations, beta %.0f, primres %.3e, mismatch %.3e" % (outer, inner, beta, res[0], res[3]))
    done = 0
    while done < inner:
        n = min(piece, inner - done)
        norms, ms, _ = b.run_fixed(n, first_major_iter=done + 1, beta=beta)
        _, _, t = b.branch_counters()
        print("  outer %d iterations %4d-%4d: %.4f ms/iteration | trips max %2d, lanes with >= 3 trips %5d, with 2 trips %5d" %
              (outer, done + 1, done + n, ms / n, t.max(), (t >= 3).sum(), (t == 2).sum()))
        done += n
        total_ms += ms
    cumul += inner
    assert np.array_equal(a.solution.u_curr, b.solution.u_curr), "replay left the solve's trajectory"
    if res[3] <= sqrt_d * outer_eps:
        break
    for m in (a, b):
        pkg._check(lib.exaadmm_update_lz(m.handle, beta), "update_lz")
    if res[2] > theta * nzp:
        beta = min(inc_c * beta, 1e24)
print("# %d inner iterations, %.2f ms of device time in the replayed pieces -> %.0f iterations/s (pieces of %d: one launch + one host visit each)"
      % (cumul, total_ms, cumul / (1e-3 * total_ms), piece))
