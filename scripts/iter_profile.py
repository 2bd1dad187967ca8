#!/usr/bin/env python3
"""GPU diagnostic: cost of an inner iteration vs its index, fixed-count mode from init_solution! (the curve that defines
bench.py's benchmark state: --prep must reach its flat part).  usage: iter_profile.py [case] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, exaadmm_loader
pkg = exaadmm_loader.load_package(); syn = exaadmm_loader.load_submodule("synthetic")
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rho = {"case1354pegase": (1e1, 1e3), "case13659pegase": (5e1, 5e3), "case_ACTIVSg70k": (3e4, 3e5), "synthetic1M": (5e1, 5e3)}[case]
g = syn.sized_like(case, seed=0, **(dict(cost="market") if case == "case_ACTIVSg70k" else {}))
env = pkg.AdmmEnv("x", rho[0], rho[1], tight_factor=0.99, verbose=0, grid=g); mod = pkg.ModelAcopf(env)
env.params.scale = 1e-5 if case == "case_ACTIVSg70k" else 1e-4; mod.sync_params()
done, chunk = 0, 20
print("# %s: device ms per iteration (chunks of %d iterations, exaadmm_run_fixed), max trips of a branch in the chunk's last iteration" % (case, chunk))
while done < n:
    norms, ms, _ = mod.run_fixed(chunk, first_major_iter=done + 1, beta=1e3)
    a, m, t = mod.branch_counters()
    print("iterations %4d-%4d: %.4f ms/iteration | primres %.4f | trips max %d, lanes with >= 3 trips %d" % (done + 1, done + chunk, ms / chunk, norms[0], t.max(), (t >= 3).sum()))
    done += chunk
