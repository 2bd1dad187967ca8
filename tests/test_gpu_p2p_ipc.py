"""-m gpu: the resident loop of a partition with peers ACROSS PROCESSES -- the form it has on a multi-GPU node (one process per GPU).

One MI355X here, so both ranks use device 0; what this covers that the thread-rank tests (test_gpu_dist_threads.py) cannot: the peers'
mailboxes are mapped through IPC handles (hipIpcGetMemHandle on the fine-grained mailbox, hipIpcOpenMemHandle in the other process),
the blobs travel through torch.distributed (gloo here), and there is NO RCCL communicator at all (RCCL refuses two ranks on one
device): the partitions live on their mailboxes alone -- resident loop, residual sums and ||z|| through the mailboxes.  The ranks'
resident kernels run on the GPU at the same time (40 wavefronts together).  Fixed-count run bitwise the single handle, the
convergence-tested solve with the single handle's iteration counts.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import hashlib, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
import exaadmm_loader
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
dm = exaadmm_loader.load_submodule("distributed")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = syn.sized_like("case1354pegase", seed=2)
rho = (1e1, 1e3)
st = dm.init_state(g, *rho)
part = dm.Partition(g, world, rank)
out = {"rank": rank, "pid": os.getpid()}

def fresh():
    be = dm.HipBackend(pkg, part, rho[0], rho[1], device_index=0)
    for name in ("v_curr", "rho"):
        be.upload(name, part.slice_global(st[name]))
    assert be.attach_native(dist, connect=False)
    return be

def gathered(model, name):
    return model.gather_owned(name)

# fixed-count run: bitwise the single handle
be = fresh()
out["p2p"], out["kind"] = bool(be.p2p), be.mod.loop_kind()
par = dict(dm.DEFAULT_PAR); par.update(outer_iterlim=3, inner_iterlim=200)
model = dm.DistributedAcopf(part, be, dist, st, par)
dist.barrier()
be.run_fixed_native(60, 1, 1e3)
out["form"], out["downgrades_fixed"] = be.mod.resident_form(), be.mod.loop_downgrades()
fixed = {n: gathered(model, n) for n in ("u_curr", "v_curr", "z_curr", "l_curr")}
# convergence-tested solve (||z|| and the loop's last sums through the mailboxes as well)
be2 = fresh()
model2 = dm.DistributedAcopf(part, be2, dist, st, par)
info = model2.solve()
out["solve"] = [info["outer"], info["cumul"], info["mismatch"]]
out["downgrades_solve"] = be2.mod.loop_downgrades()
solved = {n: gathered(model2, n) for n in ("u_curr", "v_curr")}
if rank == 0:
    env3 = pkg.AdmmEnv("single", rho[0], rho[1], verbose=0, grid=g)
    m3 = pkg.ModelAcopf(env3)
    m3.run_fixed(60, first_major_iter=1, beta=1e3)
    out["fixed_nonzero_differences"] = {n: int(np.count_nonzero(fixed[n] - getattr(m3.solution, n))) for n in fixed}
    env4, m4 = pkg.solve_acopf("syn", grid=g, rho_pq=rho[0], rho_va=rho[1], verbose=0, outer_iterlim=3, inner_iterlim=200)
    out["single"] = [m4.info.outer, m4.info.cumul, m4.info.mismatch]
    out["solve_err"] = {n: float(np.abs(solved[n] - getattr(m4.solution, n)).max() / max(1.0, np.abs(getattr(m4.solution, n)).max())) for n in solved}
dist.barrier()
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


@pytest.mark.skipif(os.environ.get("EXAADMM_DIST_P2P") == "0", reason="a switch of this run turns the mailbox loop off")
def test_two_processes_on_one_gpu_share_mailboxes_through_ipc_handles():
    if os.environ.get("EXAADMM_NO_SPEC") or os.environ.get("EXAADMM_BUS_KERNEL") == "lane" or os.environ.get("EXAADMM_NO_RESIDENT") or os.environ.get("EXAADMM_NO_REC"):
        pytest.skip("a switch of this run turns the resident loop off")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # (two processes on ONE GPU: the runtime refuses the second cooperative grid -- the TEST build's switch "resident_coop" = 0 launches plainly)
    hooks = os.path.join(ROOT, "exaadmm.jl_amd", "libexaadmm_hip_hooks.so")
    if not os.path.exists(hooks):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "exaadmm.jl_amd", "csrc"), "../libexaadmm_hip_hooks.so"])
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EXAADMM_RESIDENT_COOP="0", EXAADMM_HIP_LIB=hooks, EXAADMM_HEAVY_MAX="100", EXAADMM_HEAVY_SELECT_SMALL="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, (o[-1500:], e[-3000:])
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    r0 = [o for o in outs if o["rank"] == 0][0]
    print(outs)
    assert outs[0]["pid"] != outs[1]["pid"]
    assert all(o["p2p"] and o["kind"] == 2 and o["downgrades_fixed"] == 0 and o["downgrades_solve"] == 0 for o in outs), outs
    assert all(o["form"]["tiles"] == 1 and o["form"]["launches"] == 1 for o in outs), outs
    assert all(n == 0 for n in r0["fixed_nonzero_differences"].values()), r0
    assert r0["solve"][:2] == r0["single"][:2] and abs(r0["solve"][2] - r0["single"][2]) <= 1e-10 * r0["single"][2], r0
    assert all(e <= 1e-10 for e in r0["solve_err"].values()), r0
