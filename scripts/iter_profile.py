import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, exaadmm_loader
pkg = exaadmm_loader.load_package(); syn = exaadmm_loader.load_submodule("synthetic")
g = syn.sized_like("case13659pegase", seed=0)
env = pkg.AdmmEnv("x", 5e1, 5e3, tight_factor=0.99, verbose=0, grid=g); mod = pkg.ModelAcopf(env)
env.params.scale = 1e-4; mod.sync_params()
ms = []
for it in range(1, 161):
    n, t, tb = mod.run_fixed(1, first_major_iter=it, beta=1e3, time_branch=True)
    ms.append(t)
ms = np.array(ms)
for a, b in ((0,1),(1,5),(5,10),(10,20),(20,30),(30,40),(40,60),(60,80),(80,120),(120,160)):
    print("iterations %3d-%3d: mean %.3f ms  max %.3f" % (a+1, b, ms[a:b].mean(), ms[a:b].max()))
