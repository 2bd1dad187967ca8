#!/usr/bin/env python3
"""ONE GPU, `world` ranks as host threads over the RCCL stand-in (tests/rccl_shim): how much of the exchange does the
in-library distributed loop hide behind the interior branch solves?

    EXAADMM_RCCL_LIB=tests/rccl_shim/librccl_shim.so [RCCL_SHIM_DELAY_US=40] [EXAADMM_DIST_OVERLAP=0] \
        python scripts/dist_overlap_probe.py [world] [case] [iterations]

Workload: `world` areas of the case's size (bench.py --gpus N's weak-scaling grid), one per rank; PREP untimed iterations, then the
best of three timed runs of `iterations` fixed-count iterations (all ranks start together).  Prints one JSON line: ms per
iteration, the exchange duration sampled by the library (HIP events on the exchange stream: group + unpack), cut branches.
(All ranks share one GPU here, so absolute times are NOT those of `world` GPUs; what the numbers show is the sensitivity of
the iteration time to the exchange latency, with and without the overlap.)"""
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
case = sys.argv[2] if len(sys.argv) > 2 else "case13659pegase"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
PREP = 100
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
dm = exaadmm_loader.load_submodule("distributed")
import bench as B
params = B.CASE_PARAMS[case]
g = syn.sized_like(case, seed=0, areas=world)
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
        Stub.slots[self.rank] = int(t[0])
        Stub.barrier.wait()
        v = min(Stub.slots)
        Stub.barrier.wait()
        t[0] = v
    def all_gather_object(self, out, obj):
        Stub.slots[self.rank] = obj
        Stub.barrier.wait()
        for i in range(world): out[i] = Stub.slots[i]
        Stub.barrier.wait()


bes = [dm.HipBackend(pkg, p, params["rho_pq"], params["rho_va"], params={"scale": params["scale"]}, device_index=0) for p in parts]
for p, be in zip(parts, bes):
    be.mod.sync_params()
    for name in ("v_curr", "rho"):
        be.upload(name, p.slice_global(st[name]))
times = [[0.0] * 3 for _ in range(world)]
stats, errs = [None] * world, [None] * world


def run(r):
    try:
        assert bes[r].attach_native(Stub(r))
        lib, h = pkg.library(), bes[r].mod.handle
        if os.environ.get("P2P_DELAY_US"):   # (test build of the library: every mailbox send of the resident loop leaves this late)
            bes[r].mod.test_hook("p2p_delay_us", int(os.environ["P2P_DELAY_US"]))
        bes[r].run_fixed_native(PREP, 1, 1e3)
        pkg._check(lib.exaadmm_state_save(h), "state_save")
        for rep in range(3):
            pkg._check(lib.exaadmm_state_restore(h), "state_restore")
            Stub.barrier.wait()
            t0 = time.perf_counter()
            bes[r].run_fixed_native(iters, PREP + 1, 1e3)
            Stub.barrier.wait()
            times[r][rep] = time.perf_counter() - t0
        n, v, g_, d_, pr = C.c_int(), C.c_int(), C.c_int64(), C.c_int64(), C.c_int()
        us = C.c_double()
        pkg._check(lib.exaadmm_dist_stats(h, C.byref(n), C.byref(v), C.byref(us), C.byref(g_), C.byref(d_), C.byref(pr)), "dist_stats")
        stats[r] = {"exchange_us": us.value, "graph_batches": g_.value, "direct_batches": d_.value, "probe": pr.value,
                    "mailboxes": bool(bes[r].p2p), "loop_kind": bes[r].mod.loop_kind(), "downgrades": bes[r].mod.loop_downgrades(),
                    "resident_form": bes[r].mod.resident_form()}
    except BaseException as ex:
        errs[r] = repr(ex)
        try:
            Stub.barrier.abort()
        except Exception:
            pass


ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ths]
[t.join(600) for t in ths]
assert not any(t.is_alive() for t in ths), "a rank hung"
assert not any(errs), errs
best = min(max(times[r][rep] for r in range(world)) for rep in range(3))
print("RESULT " + json.dumps({"world": world, "case": case, "iterations": iters, "ms_per_iteration": 1e3 * best / iters,
                              "delay_us": int(os.environ.get("RCCL_SHIM_DELAY_US", "0")),
                              "overlap": os.environ.get("EXAADMM_DIST_OVERLAP", "1") != "0",
                              "cut_branches": int(sum(p.n_cut for p in parts)), "interior_blocks_rank0": None,
                              "stats": stats}))
