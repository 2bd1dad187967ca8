#!/usr/bin/env python3
"""Two processes, ONE GPU, gloo (host-staged collectives): the real HipBackend + DistributedAcopf
orchestration with real inter-process exchanges, checked against the single-handle solve.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 scripts/dist_two_ranks_one_gpu.py
(RCCL refuses two ranks on one device, so this is the closest a one-GPU box gets to the N>1 path.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import exaadmm_loader

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
pkg = exaadmm_loader.load_package()
syn = exaadmm_loader.load_submodule("synthetic")
dm = exaadmm_loader.load_submodule("distributed")
case = "case1354pegase"
g = syn.sized_like(case, seed=2)
rho = (4e2, 4e4)
part = dm.Partition(g, world, rank)
st = dm.init_state(g, *rho)
be = dm.HipBackend(pkg, part, rho[0], rho[1], device_index=0)
par = dict(dm.DEFAULT_PAR)
par.update(outer_iterlim=3, inner_iterlim=60)
model = dm.DistributedAcopf(part, be, dist, st, par)
info = model.solve()
u = model.gather_owned("u_curr")
v = model.gather_owned("v_curr")
if rank == 0:
    env, mod = pkg.solve_acopf(case, grid=g, rho_pq=rho[0], rho_va=rho[1], outer_iterlim=3, inner_iterlim=60, verbose=0)
    du = np.abs(u - mod.solution.u_curr).max()
    dv = np.abs(v - mod.solution.v_curr).max()
    ok = (info["cumul"] == mod.info.cumul) and du <= 1e-9 and dv <= 1e-9 and abs(info["mismatch"] - mod.info.mismatch) <= 1e-9 * mod.info.mismatch
    print("two ranks on one GPU: cut branches %d, cumul %d vs %d, max|du| %.2e max|dv| %.2e mismatch %.10f vs %.10f -> %s"
          % (part.n_cut, info["cumul"], mod.info.cumul, du, dv, info["mismatch"], mod.info.mismatch, "OK" if ok else "MISMATCH"), flush=True)
dist.barrier()
dist.destroy_process_group()
