"""Fixed cost of one exaadmm_run_fixed call (wall clock, host side) vs its per-iteration cost: t(K) = a + b K."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exaadmm_loader
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
g = syn.sized_like("case13659pegase", seed=0)
env = pkg.AdmmEnv("syn", 5e1, 5e3, verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
mod.run_fixed(320, first_major_iter=1, beta=1e3)
it = 321
res = {}
for K in (1, 2, 5, 20, 21, 100, 200):
    ts, dev = [], []
    for rep in range(7):
        t0 = time.perf_counter()
        _, ms, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
        ts.append((time.perf_counter() - t0) * 1e6)
        dev.append(ms * 1e3)
        it += K
    res[K] = (np.median(ts), np.median(dev))
    print("K=%4d: wall %.1f us (%.2f us/iteration), device events %.1f us (%.2f us/iteration)" % (K, res[K][0], res[K][0] / K, res[K][1], res[K][1] / K))
