#!/usr/bin/env python3
"""GPU: how many solver-loop trips do the wavefronts of one iteration execute at bench.py's state?  (calibration of
roofline.critical_path: VALU instructions per wavefront trip = SQ_INSTS_VALU per iteration of the rocprofv3 --pmc pass / wave_trips)
    python scripts/wave_trips.py [case] [prep] -> JSON: wave_trips (sum over the tiles of their largest lane count), max_trips, tiles
Run with EXAADMM_NO_HEAVY=1 for the figure that belongs to a PMC pass taken with the same switch (with heavy wavefronts the heaviest
branches' trips run on wavefronts of their own)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
import bench as B
case = sys.argv[1] if len(sys.argv) > 1 else "case13659pegase"
prep = int(sys.argv[2]) if len(sys.argv) > 2 else 320
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
params = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0)
env = pkg.AdmmEnv(case, params["rho_pq"], params["rho_va"], tight_factor=0.99, verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
env.params.scale = params["scale"]; mod.sync_params()
mod.run_fixed(prep, first_major_iter=1, beta=env.params.initial_beta)
samples = []
it = prep + 1
for _ in range(10):       # ten consecutive iterations of the steady state
    mod.run_fixed(20, first_major_iter=it, beta=env.params.initial_beta); it += 20
    t = mod.tile_trips()
    samples.append((int(t.sum()), int(t.max())))
print(json.dumps({"case": case, "state": "after %d iterations" % prep, "tiles": int(len(t)), "wave_trips_per_iteration": float(np.mean([s[0] for s in samples])),
                  "max_trips": float(np.mean([s[1] for s in samples])), "samples": samples, "loop_kind": mod.loop_kind(), "form": mod.resident_form()}))
