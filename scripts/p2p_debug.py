#!/usr/bin/env python3
"""debug harness: two thread ranks on one GPU, mailboxes mapped, a few fixed-count iterations of the resident loop with peers"""
import ctypes as C, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import exaadmm_loader
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
case = sys.argv[2] if len(sys.argv) > 2 else "case1354pegase"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters2 = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # a second run from iteration 101 (heavy wavefronts)
areas = int(os.environ.get("P2P_DEBUG_AREAS", "1"))
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
bes = [dm.HipBackend(pkg, p, params["rho_pq"], params["rho_va"], params={"scale": params["scale"]}, device_index=0) for p in parts]
for p, be in zip(parts, bes):
    be.mod.sync_params()
    for name in ("v_curr", "rho"):
        be.upload(name, p.slice_global(st[name]))
errs = [None] * world
def run(r):
    try:
        assert bes[r].attach_native(Stub(r))
        sys.stderr.write("rank %d attached p2p=%s kind=%d cut=%d\n" % (r, bes[r].p2p, bes[r].mod.loop_kind(), parts[r].n_cut))
        Stub.barrier.wait()
        res, ms = bes[r].run_fixed_native(iters, 1, 1e3)
        sys.stderr.write("rank %d done %s form=%s downgrades=%d last_error=%s\n" % (r, res, bes[r].mod.resident_form(), bes[r].mod.loop_downgrades(), pkg.library().exaadmm_last_error().decode()))
        if iters2:
            Stub.barrier.wait()
            res, ms = bes[r].run_fixed_native(iters2, 101, 1e3)
            sys.stderr.write("rank %d second run %s form=%s downgrades=%d last_error=%s\n" % (r, res, bes[r].mod.resident_form(), bes[r].mod.loop_downgrades(), pkg.library().exaadmm_last_error().decode()[:200]))
    except BaseException as ex:
        errs[r] = repr(ex)
        try: Stub.barrier.abort()
        except Exception: pass
ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ths]; [t.join(120) for t in ths]
print("errs", errs)
lib = pkg.library()
lib.exaadmm_debug_p2p_box.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64]
lib.exaadmm_debug_resident_window.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for r in range(world):
    w = (C.c_ulonglong * 4)()
    lib.exaadmm_debug_resident_window(bes[r].mod.handle, w)
    print("rank", r, "resident launch window (100 MHz ticks): start", w[0], "end", w[1], "length us", (w[1] - w[0]) / 100.0,
          "contrib", w[2] >> 32, "decided", w[2] & 0xffffffff, "stop", w[3] >> 32, "fail", w[3] & 0xffffffff, "waves", bes[r].mod.resident_form())
for r in range(world):
    buf2 = np.zeros(16 * 200)
    n = lib.exaadmm_debug_p2p_box(bes[r].mod.handle, buf2.ctypes.data_as(C.POINTER(C.c_double)), buf2.size)
    words = buf2[:16 * n].copy().view(np.uint64).reshape(n, 16)
    print("rank", r, "entries", n, "sequence numbers of the 16 words of each entry:")
    for e in list(range(3)) + list(range(n - 5, n)):
        print("  ", e, [hex(int(x) >> 32) for x in words[e]])

