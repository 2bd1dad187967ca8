#!/usr/bin/env python3
"""GPU diagnostic: time of the multi-period generator (ramp) kernel per ADMM iteration and the state of
its multipliers / penalties."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
L = pkg.library()
case = sys.argv[1] if len(sys.argv) > 1 else "case1354pegase"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = syn.sized_like(case, seed=0)
lvl = 1.0 + 0.01 * np.sin(np.arange(T) * 2 * np.pi / T)
rho = (5e1, 5e3) if "13659" in case else (1e1, 1e3)
env = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g)
pd = np.stack([env.data["Pd"] * a for a in lvl], 1); qd = np.stack([env.data["Qd"] * a for a in lvl], 1)
mod = pkg.ModelMpacopf(env, (pd, qd), end_period=T)
env.params.beta = 1e3
mod.sync_params()
for it in range(1, 41):
    mod.info.inner = it
    pkg.admm_inner_prestep(env, mod)
    t0 = time.perf_counter()
    pkg._check(L.exaadmm_mp_update_x_gen(mod.handle, it), "gen")
    t1 = time.perf_counter()
    pkg._check(L.exaadmm_update_x_line(mod.handle, it), "line")
    t2 = time.perf_counter()
    pkg.admm_update_xbar(env, mod); pkg.admm_update_z(env, mod); pkg.admm_update_l(env, mod); pkg.admm_update_residual(env, mod)
    if it in (1, 2, 3, 5, 10, 20, 30, 40):
        xi = mod.download_ramp("xi")[mod.ngen:]; mu = mod.download_ramp("mu")[mod.ngen:]
        print("it %2d gen %.3f ms line %.3f ms | xi: %s | |mu| max %.3g | primres %.3g" % (
            it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), dict(zip(*[a.tolist() for a in np.unique(xi, return_counts=True)])), np.abs(mu).max(), mod.info.primres), flush=True)
