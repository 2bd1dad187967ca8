#!/usr/bin/env python3
"""GPU stress of the resident loop WITH PEERS (mailboxes): `world` thread ranks on one GPU, REPS repetitions of the same sequence of
fixed-count runs (lengths that cross the 64th iteration: launches with and without heavy wavefronts) from the same start, every
repetition's owned u / v / z / l compared BITWISE with the single handle's.  Counts mismatching repetitions and downgrades.
    EXAADMM_RCCL_LIB=tests/rccl_shim/librccl_shim.so GPU_MAX_HW_QUEUES=16 EXAADMM_RESIDENT_COOP=0 python scripts/stress_p2p.py [world] [case] [areas] [reps]"""
import ctypes as C, hashlib, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
case = sys.argv[2] if len(sys.argv) > 2 else "case1354pegase"
areas = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
PLAN = (1, 2, 13, 60, 33, 5)      # 114 iterations; from the 4th run on the launches start beyond iteration 64
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
dm = exaadmm_loader.load_submodule("distributed")
import bench as B
params = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0, areas=areas)
st = dm.init_state(g, params["rho_pq"], params["rho_va"])
parts = [dm.Partition(g, world, r) for r in range(world)]
buf = C.create_string_buffer(128)
pkg._check(pkg.library().exaadmm_dist_unique_id(buf, 128), "unique_id")
class Stub:
    class ReduceOp: MIN = "min"
    barrier = threading.Barrier(world)
    slots = [None] * world
    def __init__(self, rank): self.rank = rank
    def get_world_size(self): return world
    def get_rank(self): return self.rank
    def get_backend(self): return "threads"
    def broadcast_object_list(self, lst, src=0): lst[0] = buf.raw
    def all_reduce(self, t, op=None):
        Stub.slots[self.rank] = int(t[0]); Stub.barrier.wait(); v = min(Stub.slots); Stub.barrier.wait(); t[0] = v
    def all_gather_object(self, out, obj):
        Stub.slots[self.rank] = obj; Stub.barrier.wait()
        for i in range(world): out[i] = Stub.slots[i]
        Stub.barrier.wait()
# the single handle's result
env1 = pkg.AdmmEnv(case, params["rho_pq"], params["rho_va"], verbose=0, grid=g)
m1 = pkg.ModelAcopf(env1); env1.params.scale = params["scale"]; m1.sync_params()
it = 1
for K in PLAN:
    m1.run_fixed(K, first_major_iter=it, beta=1e3); it += K
ref = {n: getattr(m1.solution, n) for n in ("u_curr", "v_curr", "z_curr", "l_curr")}
m1.close()
bes = [dm.HipBackend(pkg, p, params["rho_pq"], params["rho_va"], params={"scale": params["scale"]}, device_index=0) for p in parts]
for be in bes: be.mod.sync_params()
bad, downs, errs, t_run = [0] * world, [0] * world, [None] * world, [0.0] * world
def run(r):
    try:
        assert bes[r].attach_native(Stub(r)) and bes[r].p2p and bes[r].mod.loop_kind() == 2
        if os.environ.get("EXAADMM_TEST_P2P_PATIENT_MS"):   # (test build: thread ranks on one GPU can have a kernel held up behind another rank's -- then 3 s, not 60)
            bes[r].mod.test_hook("p2p_patient_ms", int(os.environ["EXAADMM_TEST_P2P_PATIENT_MS"]))
        p, lib, h = parts[r], pkg.library(), bes[r].mod.handle
        gi = p.local_index_map()[:p.nvar_owned]
        for rep in range(reps):
            pkg._check(lib.exaadmm_init_solution(h, params["rho_pq"], params["rho_va"]), "init")
            for name in ("v_curr", "rho"):
                bes[r].upload(name, p.slice_global(st[name]))
            bes[r].mod.membuf = bes[r].mod.membuf * 0 + np.where(np.arange(31)[:, None] == 28, bes[r].mod.membuf, 0.0)   # multipliers / penalty back to zero, rateA kept
            Stub.barrier.wait()
            t0 = time.perf_counter(); it = 1
            for K in PLAN:
                bes[r].run_fixed_native(K, it, 1e3); it += K
            t_run[r] += time.perf_counter() - t0
            ok = all(np.array_equal(np.asarray(bes[r].download(n))[:p.nvar_owned], ref[n][gi]) for n in ref)
            bad[r] += 0 if ok else 1
        downs[r] = bes[r].mod.loop_downgrades()
    except BaseException as ex:
        errs[r] = repr(ex)
        try: Stub.barrier.abort()
        except Exception: pass
ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ths]; [t.join(3000) for t in ths]
print("RESULT " + json.dumps({"world": world, "case": case, "areas": areas, "repetitions": reps, "iterations_per_repetition": sum(PLAN),
                              "mismatching_repetitions_per_rank": bad, "downgrades_per_rank": downs, "errors": errs,
                              "seconds_in_the_runs_rank0": round(t_run[0], 2), "cut_branches": int(sum(p.n_cut for p in parts))}))
