"""-m gpu: the distributed inner loop INSIDE the library (csrc/dist_native.h) with 2 and 3 ranks on a one-GPU box.

Real RCCL refuses two ranks on one device, so the ranks are host threads of one process, each driving its own handle
(= its own partition and stream), and the nine RCCL entry points the library binds at run time are provided by a
test-only in-process stand-in (tests/rccl_shim/rccl_shim.cpp, selected with EXAADMM_RCCL_LIB).  What this executes
that nothing else can on one GPU: the C++ loop with real peers -- halo gather / grouped send + receive / scatter in
both directions, the all-reduce of the residual sums, identical break decisions on every rank, eight iterations per
host visit.  Must reproduce the fused single-GPU solve: same outer / cumulative iteration counts, u / v / z to 1e-10.
Runs in a subprocess (the RCCL binding is per process)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys, threading
sys.path.insert(0, %(root)r)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
dm = exaadmm_loader.load_submodule("distributed")
import ctypes as C
world = %(world)d
case = %(case)r
if case:
    g = pkg.opf_loaddata(case)            # the reference's 118-bus anchor case, cut into `world` partitions
    rho, par_over = (4e2, 4e4), dict(outer_iterlim=25, outer_eps=2e-5)
else:
    g = syn.sized_like("case1354pegase", seed=2)
    rho, par_over = (1e1, 1e3), dict(outer_iterlim=3, inner_iterlim=200)
st = dm.init_state(g, *rho)
parts = [dm.Partition(g, world, r) for r in range(world)]
assert sum(p.n_cut for p in parts) > 0
buf = C.create_string_buffer(128)
pkg._check(pkg.library().exaadmm_dist_unique_id(buf, 128), "unique_id")
assert b"rccl_shim" in pkg.library().exaadmm_dist_rccl_path()

class Stub:
    def __init__(self, rank): self.rank = rank
    def get_world_size(self): return world
    def get_rank(self): return self.rank
    def get_backend(self): return "nccl"
    def broadcast_object_list(self, lst, src=0): lst[0] = buf.raw

bes = [dm.HipBackend(pkg, p, rho[0], rho[1], device_index=0) for p in parts]
par = dict(dm.DEFAULT_PAR); par.update(par_over)
models = [dm.DistributedAcopf(p, be, Stub(r), st, par) for r, (p, be) in enumerate(zip(parts, bes))]
infos, errs = [None] * world, [None] * world
def run(r):
    try:
        assert bes[r].attach_native(Stub(r))
        e = C.c_double(1.0)
        pkg._check(pkg.library().exaadmm_dist_selftest(bes[r].mod.handle, 1000, C.byref(e)), "selftest")
        assert e.value == 0.0, e.value
        infos[r] = models[r].solve()
    except BaseException as ex:
        errs[r] = repr(ex)
ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ths]; [t.join(180) for t in ths]
assert not any(t.is_alive() for t in ths), "a rank hung"
assert not any(errs), errs
env, mod = pkg.solve_acopf("syn", grid=g, rho_pq=rho[0], rho_va=rho[1], verbose=0, **par_over)
out = {"cut": int(sum(p.n_cut for p in parts)), "cumul": [i["cumul"] for i in infos], "outer": [i["outer"] for i in infos],
       "mismatch": [i["mismatch"] for i in infos], "single": [mod.info.outer, mod.info.cumul, mod.info.mismatch], "err": {}}
for name in ("u_curr", "v_curr", "z_curr", "l_curr"):
    glob = np.zeros(parts[0].nvar_global)
    for p, be in zip(parts, bes):
        glob[p.local_index_map()[:p.nvar_owned]] = be.download(name)[:p.nvar_owned]
    one = getattr(mod.solution, name)
    out["err"][name] = float(np.abs(glob - one).max() / max(1.0, np.abs(one).max()))
# fixed-count mode (the benchmark path): the partitioned loop must be BITWISE the single-handle loop -- every kernel sees
# the same inputs in the same order, only the residual sums are added in a different order (and are not fed back)
bes2 = [dm.HipBackend(pkg, p, rho[0], rho[1], device_index=0) for p in parts]
for p, be in zip(parts, bes2):
    for name in ("v_curr", "rho"):
        be.upload(name, p.slice_global(st[name]))
errs = [None] * world
def run2(r):
    try:
        assert bes2[r].attach_native(Stub(r))
        bes2[r].run_fixed_native(60, 1, 1e3)
    except BaseException as ex:
        errs[r] = repr(ex)
ths = [threading.Thread(target=run2, args=(r,)) for r in range(world)]
[t.start() for t in ths]; [t.join(180) for t in ths]
assert not any(t.is_alive() for t in ths), "a rank hung"
assert not any(errs), errs
env3 = pkg.AdmmEnv("single", rho[0], rho[1], verbose=0, grid=g)
m3 = pkg.ModelAcopf(env3)
m3.run_fixed(60, first_major_iter=1, beta=1e3)
out["fixed_nonzero_differences"] = {}
for name in ("u_curr", "v_curr", "z_curr", "l_curr"):
    glob = np.zeros(parts[0].nvar_global)
    for p, be in zip(parts, bes2):
        glob[p.local_index_map()[:p.nvar_owned]] = be.download(name)[:p.nvar_owned]
    out["fixed_nonzero_differences"][name] = int(np.count_nonzero(glob - getattr(m3.solution, name)))
print("RESULT " + json.dumps(out))
'''


def _shim():
    so = os.path.join(ROOT, "tests", "rccl_shim", "librccl_shim.so")
    src = os.path.join(ROOT, "tests", "rccl_shim", "rccl_shim.cpp")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-fPIC", "-shared", "-I/opt/rocm/include", src, "-o", so])
    return so


def _run_worker(world, case):
    env = dict(os.environ, EXAADMM_RCCL_LIB=_shim())
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT, "world": world, "case": case}], env=env,
                       capture_output=True, text=True, timeout=480)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


@pytest.mark.parametrize("world", [2, 3])
def test_library_distributed_loop_with_several_ranks_on_one_gpu(world):
    out = _run_worker(world, "")
    assert out["cut"] > 0
    assert all(c == out["single"][1] for c in out["cumul"]) and all(o == out["single"][0] for o in out["outer"]), out
    assert all(abs(m - out["single"][2]) <= 1e-10 * out["single"][2] for m in out["mismatch"]), out
    assert all(e <= 1e-10 for e in out["err"].values()), out
    assert all(n == 0 for n in out["fixed_nonzero_differences"].values()), out


def test_library_distributed_loop_reproduces_the_case118_anchor():
    """The reference's 118-bus anchor (acopf_update_cpu.jl:174-179: 20 outer / 1232 inner iterations) from the
    in-library distributed loop with the system cut into 3 partitions."""
    out = _run_worker(3, os.path.join(ROOT, "tests", "golden", "case118.m"))
    assert out["cut"] > 0
    assert out["cumul"] == [1232] * 3 and out["outer"] == [20] * 3, out
    assert out["single"][:2] == [20, 1232], out
    assert all(e <= 1e-10 for e in out["err"].values()), out
    assert all(n == 0 for n in out["fixed_nonzero_differences"].values()), out
