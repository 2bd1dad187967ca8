"""world_size-2 (and 3) gloo tests of the multi-GPU orchestration: partitioning, cut branches replicated and
solved on both owners, the ONE halo exchange per iteration (v entries of the cut branches' local ends),
residual all-reduce and device-style loop control must reproduce the single-process solve of the global grid."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import exaadmm_loader


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


GRID = dict(ngen=14, nline=170, nbus=110, seed=5, locality=12)
RHO = (4e2, 4e4)
PAR = dict(outer_iterlim=2, inner_iterlim=25)


def _worker(rank, world, port, outdir, check_every=1, case=None, par_over=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dm = exaadmm_loader.load_submodule("distributed")
        syn = exaadmm_loader.load_submodule("synthetic")
        from dist_oracle_backend import OracleBackend
        if case is None:
            grid = syn.synthetic_grid(GRID["ngen"], GRID["nline"], GRID["nbus"], seed=GRID["seed"], locality=GRID["locality"])
        else:
            from oracle import oracle as O
            grid = O.grid_from_matpower(case)
        part = dm.Partition(grid, world, rank)
        st = dm.init_state(grid, *RHO)
        be = OracleBackend(part, *RHO)
        par = dict(dm.DEFAULT_PAR)
        par.update(PAR if par_over is None else par_over)
        model = dm.DistributedAcopf(part, be, dist, st, par, check_every=check_every)
        info = model.solve()
        u = model.gather_owned("u_curr")
        v = model.gather_owned("v_curr")
        z = model.gather_owned("z_curr")
        cut = torch.tensor([float(part.n_cut)])
        dist.all_reduce(cut)
        if rank == 0:
            np.savez(os.path.join(outdir, "out.npz"), u=u, v=v, z=z, cumul=info["cumul"], outer=info["outer"],
                     primres=info["primres"], mismatch=info["mismatch"], cut=float(cut[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_solve_matches_single_process(world, tmp_path, oracle):
    syn = exaadmm_loader.load_submodule("synthetic")
    grid = syn.synthetic_grid(GRID["ngen"], GRID["nline"], GRID["nbus"], seed=GRID["seed"], locality=GRID["locality"])
    ref = oracle.solve_acopf("x", grid=grid, rho_pq=RHO[0], rho_va=RHO[1], **PAR)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = np.load(os.path.join(str(tmp_path), "out.npz"))
    assert out["cut"] > 0                      # the test really exercises the halo
    assert int(out["cumul"]) == ref.info.cumul and int(out["outer"]) == ref.info.outer
    for name, key in (("u_curr", "u"), ("v_curr", "v"), ("z_curr", "z")):
        a, b = out[key], ref.sol[name]
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), name
    assert np.isclose(float(out["primres"]), ref.info.primres, rtol=1e-10)
    assert np.isclose(float(out["mismatch"]), ref.info.mismatch, rtol=1e-10)


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_solve_reproduces_the_case118_anchor(world, tmp_path):
    """The reference's second anchor (test/algorithms/acopf_update_cpu.jl:174-179: 20 outer / 1232 inner iterations,
    objective 129645.676) with the 118-bus system cut into 2 and 3 partitions: halo exchange, replicated break
    decisions and the all-reduced residuals leave the trajectory where the single-process reference has it."""
    from conftest import CASE118
    from oracle import oracle as O
    anchor = json.load(open(os.path.join(os.path.dirname(CASE118), "case9_iter1.json")))["solve_case118"]
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), 1, CASE118, dict(outer_iterlim=25, outer_eps=2e-5)),
             nprocs=world, join=True)
    out = np.load(os.path.join(str(tmp_path), "out.npz"))
    assert out["cut"] > 0
    assert int(out["cumul"]) == anchor["cumul"] and int(out["outer"]) == anchor["outer"]
    grid = O.grid_from_matpower(CASE118)
    pg = out["u"][0:2 * grid["ngen"]:2] * grid["baseMVA"]
    # (admm_poststep, acopf_admm_prepoststep_cpu.jl:30-46: raw cost coefficients on MW)
    raw = O.read_matpower(CASE118)["mpc.gencost"]
    c2, c1, c0 = (np.array([r[k] for r in raw]) for k in (4, 5, 6))
    objval = float(np.sum(c2 * pg ** 2 + c1 * pg + c0))
    assert abs(objval - anchor["objval"]) <= 1e-6 * anchor["objval"]


def test_partition_invariants():
    dm = exaadmm_loader.load_submodule("distributed")
    syn = exaadmm_loader.load_submodule("synthetic")
    grid = syn.synthetic_grid(40, 600, 400, seed=1, locality=16)
    world = 4
    parts = [dm.Partition(grid, world, r) for r in range(world)]
    # every branch is native on exactly one rank; every generator / bus owned once
    assert sorted(np.concatenate([p.native for p in parts]).tolist()) == list(range(600))
    assert sorted(np.concatenate([p.gens for p in parts]).tolist()) == list(range(40))
    assert sorted(np.concatenate([p.buses for p in parts]).tolist()) == list(range(400))
    for a in parts:
        assert len(a.cut[a.rank]) == 0
        for b in parts:
            if a.rank != b.rank:
                # both owners of a cut branch list it, in the same (global id) order; 4 doubles per branch each way
                assert np.array_equal(a.cut[b.rank], b.cut[a.rank])
                assert a.plan_send_v[2][b.rank] == b.plan_recv_v[2][a.rank] == 4 * len(a.cut[b.rank])
        # what goes out are the v entries of the LOCAL ends, what comes in those of the REMOTE ends: disjoint, and
        # together all 8 entries of every cut branch
        sent, recv = set(a.plan_send_v[1].tolist()), set(a.plan_recv_v[1].tolist())
        assert not (sent & recv) and len(sent) == len(recv) == 4 * len(a.cut_local)
        assert sent | recv == set((2 * a.Gl + 8 * a.cut_local[:, None] + np.arange(8)[None, :]).ravel().tolist())
    # owned entries of all ranks tile the global vector exactly once
    cover = np.zeros(2 * 40 + 8 * 600, dtype=int)
    for p in parts:
        cover[p.local_index_map()[:p.nvar_owned]] += 1
    assert (cover == 1).all()
    # each local CSR only references local branches incident to owned buses
    for p in parts:
        g = p.grid
        assert g["FrStart"][-1] == p.n_native
        assert g["ToStart"][-1] <= p.Ll


def test_bfs_order_is_a_permutation():
    dm = exaadmm_loader.load_submodule("distributed")
    syn = exaadmm_loader.load_submodule("synthetic")
    grid = syn.synthetic_grid(5, 80, 50, seed=2, locality=6)
    order = dm.bfs_bus_order(grid)
    assert sorted(order.tolist()) == list(range(50))


def test_relaxed_check_every_runs_at_most_k_minus_1_iterations_longer(tmp_path, oracle):
    """opt-in relaxation: residual all-reduce + break test every 4th inner iteration"""
    syn = exaadmm_loader.load_submodule("synthetic")
    grid = syn.synthetic_grid(GRID["ngen"], GRID["nline"], GRID["nbus"], seed=GRID["seed"], locality=GRID["locality"])
    ref = oracle.solve_acopf("x", grid=grid, rho_pq=RHO[0], rho_va=RHO[1], **PAR)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), 4), nprocs=2, join=True)
    out = np.load(os.path.join(str(tmp_path), "out.npz"))
    assert int(out["outer"]) == ref.info.outer
    assert ref.info.cumul <= int(out["cumul"]) <= ref.info.cumul + 3 * ref.info.outer
    assert np.isfinite(out["u"]).all() and np.abs(out["u"] - ref.sol["u_curr"]).max() < 0.5


def test_default_partition_keeps_the_cut_small_whatever_the_bus_numbering():
    """Real case files number their buses arbitrarily: cutting the loader's order into ranges would cut almost
    every branch.  The default partition cuts a breadth-first order instead (Partition(reorder=True))."""
    dm = exaadmm_loader.load_submodule("distributed")
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.synthetic_grid(60, 2000, 1300, seed=9, locality=16)
    rng = np.random.default_rng(0)
    perm = rng.permutation(1300)                       # shuffle the bus labels: the banded structure is hidden
    inv = np.argsort(perm)
    h = dict(g)
    bb = perm[np.asarray(g["brBusIdx"])]
    h["brBusIdx"] = bb
    for k in ("Pd", "Qd", "Vmin", "Vmax", "YshR", "YshI"):
        h[k] = np.asarray(g[k])[inv]
    gb = perm[np.asarray(g["gen_bus"])]

    def csr(owner, n):
        order = np.argsort(owner, kind="stable").astype(np.int64)
        start = np.zeros(n + 1, dtype=np.int64)
        np.add.at(start, owner + 1, 1)
        return np.cumsum(start), order

    bb2 = bb.reshape(-1, 2)
    h["FrStart"], h["FrIdx"] = csr(bb2[:, 0], 1300)
    h["ToStart"], h["ToIdx"] = csr(bb2[:, 1], 1300)
    h["GenStart"], h["GenIdx"] = csr(gb, 1300)
    h["gen_bus"] = gb
    world = 8
    cut_bfs = sum(dm.Partition(h, world, r).n_cut for r in range(world))
    cut_raw = sum(dm.Partition(h, world, r, reorder=False).n_cut for r in range(world))
    cut_band = sum(dm.Partition(g, world, r, reorder=False).n_cut for r in range(world))   # the generator's own (banded) numbering
    assert cut_raw > 0.8 * 2000 * (1 - 1 / world)       # shuffled labels cut into ranges: nearly every branch is cut
    assert cut_bfs <= 0.12 * 2000, (cut_bfs, cut_raw, cut_band)
    assert cut_bfs <= 3 * max(cut_band, 1)
    # every branch is native on exactly one rank, every cut branch is a ghost on exactly one other
    parts = [dm.Partition(h, world, r) for r in range(world)]
    assert sum(p.n_native for p in parts) == 2000 and sum(p.n_ghost for p in parts) == cut_bfs


def test_weak_scaling_workload_is_n_areas_with_few_tie_lines():
    """bench.py --gpus N (--scaling weak): ONE banded grid of N areas of the case's size, cut N ways along a breadth-first bus
    order -- every rank gets about one area, the cut branches are the tie lines (north_star's target configuration is this
    with --case case_ACTIVSg70k; here at 1 354-bus size, 4 areas)."""
    dm = exaadmm_loader.load_submodule("distributed")
    syn = exaadmm_loader.load_submodule("synthetic")
    g1, l1, b1 = syn.CASE_SIZES["case1354pegase"]
    world = 4
    grid = syn.sized_like("case1354pegase", seed=0, areas=world)
    assert (grid["ngen"], grid["nline"], grid["nbus"]) == (world * g1, world * l1, world * b1)
    parts = [dm.Partition(grid, world, r) for r in range(world)]
    assert sum(p.n_native for p in parts) == world * l1
    for p in parts:
        assert abs(p.Bl - b1) <= 1 and abs(p.n_native - l1) <= 0.1 * l1            # one area's worth of work per rank
        assert p.grid["nline_owned"] == p.n_native and p.Ll == p.n_native + p.n_ghost
    cut = sum(p.n_cut for p in parts)
    assert 0 < cut <= 0.03 * world * l1, cut                                        # tie lines only
    # the ACTIVSg70k-sized case carries its own generator options (market-scale costs, which the README's rho is tuned to)
    assert syn.CASE_KW["case_ACTIVSg70k"] == {"cost": "market"}


def test_north_star_scaling_workload_of_70k_sized_areas_can_be_generated_and_cut():
    """bench.py --gpus N --case case_ACTIVSg70k: N areas of 70 000 buses in one band.  The DC angles of such a band drift to the ACOPF's
    +-2 pi bounds unless the generator removes the long-range trend (exaadmm.jl_amd/synthetic.py); the two-area grid must build, keep
    |theta| < 3 and small angle differences, and be cut into two partitions with a handful of tie lines."""
    syn = exaadmm_loader.load_submodule("synthetic")
    dm = exaadmm_loader.load_submodule("distributed")
    g = syn.sized_like("case_ACTIVSg70k", seed=0, areas=2)
    assert (int(g["nbus"]), int(g["nline"])) == (140000, 176414)
    lo, hi = np.asarray(g["FrVaBound"]).reshape(-1, 2).T
    assert np.all(lo <= hi)
    parts = [dm.Partition(g, 2, r) for r in range(2)]
    ncut = sum(p.n_cut for p in parts)
    assert 0 < ncut <= 200, ncut
    assert abs(parts[0].Bl - parts[1].Bl) <= 1 and all(p.n_native > 80000 for p in parts)
