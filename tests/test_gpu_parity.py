"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle
on the same inputs, against the reference's golden vectors, and - at benchmark sizes -
through size-independent properties.  Tolerances are written next to each assertion."""
import json
import os

import numpy as np
import pytest

from conftest import CASE9, GOLDEN
import exaadmm_loader

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(GOLDEN, "case9_iter1.json")))


@pytest.fixture(scope="module")
def pkg():
    p = exaadmm_loader.load_package()
    p.library()  # raises if the HIP extension is missing
    return p


@pytest.fixture(scope="module")
def syn():
    return exaadmm_loader.load_submodule("synthetic")


def _one_iter_stepwise(pkg, env, mod, inner=1):
    mod.info.inner = inner - 1
    pkg.admm_increment_inner(env, mod)
    pkg.admm_inner_prestep(env, mod)
    pkg.admm_update_x(env, mod)
    pkg.admm_update_xbar(env, mod)
    pkg.admm_update_z(env, mod)
    pkg.admm_update_l(env, mod)
    pkg.admm_update_residual(env, mod)


def test_case9_one_iteration_reference_known_answers(pkg):
    """Same sequence and vectors as test/algorithms/acopf_update_gpu.jl, atol 1e-6."""
    env = pkg.AdmmEnv(CASE9, 4e2, 4e4, verbose=0)
    mod = pkg.ModelAcopf(env)
    env.params.scale, env.params.initial_beta, env.params.beta = 1e-4, 1e3, 1e3
    pkg.admm_increment_outer(env, mod)
    pkg.admm_outer_prestep(env, mod)
    pkg.admm_increment_reset_inner(env, mod)
    pkg.admm_increment_inner(env, mod)
    pkg.admm_inner_prestep(env, mod)
    sol = mod.solution
    atol = G["atol"] + 1e-9   # 1e-6 of the reference test; its literals are truncated at 6 decimals
    pkg.admm_update_x(env, mod)
    u = sol.u_curr
    assert np.abs(u[:6] - G["U_GEN"]).max() <= atol
    assert np.abs(u[6:] - G["U_BR"]).max() <= atol
    pkg.admm_update_xbar(env, mod)
    v = sol.v_curr
    assert np.abs(v[:6] - G["V_GEN"]).max() <= atol
    assert np.abs(v[6:] - G["V_BR"]).max() <= atol
    pkg.admm_update_z(env, mod)
    z = sol.z_curr
    assert np.abs(z[:6] - G["Z_GEN"]).max() <= atol
    assert np.abs(z[6:] - G["Z_BR"]).max() <= atol
    pkg.admm_update_l(env, mod)
    l = sol.l_curr
    assert np.abs(l[:6] - G["L_GEN"]).max() <= atol + 1e-6   # literals truncated at 6 decimals
    assert np.abs(l[6:] - G["L_BR"]).max() <= atol + 1e-6
    pkg.admm_update_residual(env, mod)
    assert np.abs(sol.rp - (sol.u_curr - sol.v_curr + sol.z_curr)).max() <= atol
    assert np.abs(sol.rd - (sol.z_curr - sol.z_prev)).max() <= atol
    assert np.abs(sol.Ax_plus_By - (sol.u_curr - sol.v_curr)).max() <= atol
    lz_prev = sol.lz
    pkg.admm_update_lz(env, mod)
    assert np.abs(sol.lz - (lz_prev + env.params.beta * sol.z_curr)).max() <= atol


@pytest.mark.parametrize("fused", [True, False])
def test_case9_full_solve_anchor(pkg, fused):
    """Solved / outer 20 / cumul 705 / objval 5303.435 (acopf_update_gpu.jl:190-194)."""
    a = G["solve_case9"]
    env, mod = pkg.solve_acopf(CASE9, outer_iterlim=a["outer_iterlim"], rho_pq=4e2, rho_va=4e4,
                               outer_eps=a["outer_eps"], verbose=0, fused=fused)
    assert mod.info.status == "Solved"
    assert mod.info.outer == a["outer"]
    assert mod.info.cumul == a["cumul"]
    assert abs(mod.info.objval - a["objval"]) <= a["objval_atol"]


def test_case9_trajectory_matches_oracle(pkg, oracle, case9_grid):
    """10 step-wise iterations: every state vector within 1e-8 (abs, values are O(1..100))."""
    env = pkg.AdmmEnv(CASE9, 4e2, 4e4, verbose=0)
    mod = pkg.ModelAcopf(env)
    ref = oracle.OracleModel(case9_grid, 4e2, 4e4)
    mod.info.outer = 1
    for it in range(1, 11):
        _one_iter_stepwise(pkg, env, mod, it)
        ref.inner_prestep(); ref.update_x(it); ref.update_xbar(); ref.update_z(); ref.update_l()
        res = ref.update_residual()
        for name in ("u_curr", "v_curr", "z_curr", "l_curr"):
            scale = max(1.0, np.abs(ref.sol[name]).max())
            assert np.abs(getattr(mod.solution, name) - ref.sol[name]).max() <= 1e-8 * scale, (it, name)
        got = (mod.info.primres, mod.info.dualres, mod.info.norm_z_curr, mod.info.mismatch)
        assert np.allclose(got, res, rtol=1e-9, atol=1e-12)
    mb, mbr = mod.membuf, ref.membuf.T
    assert np.allclose(mb[24:27], mbr[24:27], rtol=1e-7, atol=1e-9)
    assert np.array_equal(mb[28], mbr[28])


@pytest.mark.parametrize("case", ["case118", "case1354pegase"])
def test_synthetic_fused_loop_matches_oracle(pkg, oracle, syn, case):
    """Fused device-resident inner loop vs oracle on size-matched synthetic grids:
    same iteration count, u/v to 1e-6 relative (FP64 with different summation orders and FMA)."""
    g = syn.sized_like(case, seed=3)
    rho = (4e2, 4e4) if case == "case118" else (1e1, 1e3)
    kw = dict(outer_iterlim=2, inner_iterlim=40, rho_pq=rho[0], rho_va=rho[1], scale=1e-4)
    env, mod = pkg.solve_acopf(case, grid=g, verbose=0, **kw)
    ref = oracle.solve_acopf(case, grid=g, **kw)
    assert mod.info.cumul == ref.info.cumul
    assert mod.info.outer == ref.info.outer
    for name in ("u_curr", "v_curr"):
        a, b = getattr(mod.solution, name), ref.sol[name]
        assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), name
    assert abs(mod.info.objval - ref.info.objval) <= 1e-6 * abs(ref.info.objval)
    assert np.isclose(mod.info.primres, ref.info.primres, rtol=1e-6)


def test_ragged_and_edge_grids(pkg, oracle, syn):
    """sizes not multiples of the 64-lane padding, a generator-free bus majority, and a bus with
    many incident branches"""
    for (G_, L_, B_) in [(1, 63, 40), (2, 65, 64), (5, 129, 100)]:
        g = syn.synthetic_grid(G_, L_, B_, seed=G_ + L_, locality=8)
        kw = dict(outer_iterlim=1, inner_iterlim=12, rho_pq=4e2, rho_va=4e4)
        env, mod = pkg.solve_acopf("x", grid=g, verbose=0, **kw)
        ref = oracle.solve_acopf("x", grid=g, **kw)
        assert mod.info.cumul == ref.info.cumul
        b = ref.sol["u_curr"]
        assert np.abs(mod.solution.u_curr - b).max() <= 1e-6 * max(1.0, np.abs(b).max())


def test_upload_download_roundtrip_and_membuf(pkg):
    env = pkg.AdmmEnv(CASE9, 4e2, 4e4, verbose=0)
    mod = pkg.ModelAcopf(env)
    rng = np.random.default_rng(0)
    for name in ("u_curr", "v_curr", "l_curr", "z_curr", "lz", "rho"):
        x = rng.normal(size=mod.nvar)
        setattr(mod.solution, name, x)
        assert np.array_equal(getattr(mod.solution, name), x)
    mb = np.zeros((31, mod.nline))
    mb[24], mb[25], mb[26], mb[28] = rng.normal(size=(4, mod.nline))
    mod.membuf = mb
    assert np.array_equal(mod.membuf, mb)


def test_properties_at_benchmark_size(pkg, syn):
    """case13659pegase-sized grid: determinism (bitwise) of the fused loop, residual identities,
    v consensus (all branch ends at a bus see the same w, theta), flows consistent with (w, theta)."""
    g = syn.sized_like("case13659pegase")
    outs = []
    for _ in range(2):
        env = pkg.AdmmEnv("syn", 5e1, 5e3, verbose=0, grid=g)
        mod = pkg.ModelAcopf(env)
        env.params.scale = 1e-4
        mod.sync_params()
        norms, ms, _ = mod.run_fixed(5, first_major_iter=1, beta=1e3)
        outs.append((mod.solution.u_curr, mod.solution.v_curr, norms))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])
    u, v = outs[0][0], outs[0][1]
    sol = mod.solution
    z, zp = sol.z_curr, sol.z_prev
    assert np.allclose(sol.rp, u - v + z, atol=1e-12)
    assert np.allclose(sol.rd, z - zp, atol=1e-12)
    nrm = np.sqrt([np.sum((u - v + z) ** 2), np.sum((z - zp) ** 2), np.sum(z ** 2), np.sum((u - v) ** 2)])
    assert np.allclose(outs[0][2], nrm, rtol=1e-10)
    G_, L_ = g["ngen"], g["nline"]
    vb = v[2 * G_:].reshape(L_, 8)
    bb = g["brBusIdx"].reshape(L_, 2)
    ends_bus = np.concatenate([bb[:, 0], bb[:, 1]])
    for col_fr, col_to in ((4, 5), (6, 7)):   # w and theta consensus
        ends_val = np.concatenate([vb[:, col_fr], vb[:, col_to]])
        at_bus = np.full(g["nbus"], np.nan)
        at_bus[ends_bus] = ends_val
        assert np.array_equal(at_bus[ends_bus], ends_val)
    ub = u[2 * G_:].reshape(L_, 8)
    vi, vj, th = np.sqrt(ub[:, 4]), np.sqrt(ub[:, 5]), ub[:, 6] - ub[:, 7]
    pij = g["YffR"] * vi ** 2 + g["YftR"] * vi * vj * np.cos(th) + g["YftI"] * vi * vj * np.sin(th)
    assert np.allclose(ub[:, 0], pij, rtol=1e-9, atol=1e-9)
    assert (ub[:, 4] >= 0.81 - 1e-12).all() and (ub[:, 4] <= 1.21 + 1e-12).all()


@pytest.mark.parametrize("world", [2, 3])
def test_two_partitions_on_one_gpu_match_single_handle(pkg, syn, oracle, world):
    """Multi-GPU helpers of the C ABI on real hardware (the step-wise twin of the in-library loop): two (three)
    partitions = handles of one grid driven in lockstep on the same GPU, cut branches solved on both owners, the ONE
    exchange per iteration (v entries of the cut branches' local ends) emulated by device-to-device copies in the
    all_to_all layout; must reproduce the single-handle trajectory (1e-10) and the oracle."""
    import torch
    dm = exaadmm_loader.load_submodule("distributed")
    g = syn.sized_like("case1354pegase", seed=2)
    rho = (1e1, 1e3)
    st = dm.init_state(g, *rho)
    parts = [dm.Partition(g, world, r) for r in range(world)]
    assert sum(p.n_cut for p in parts) > 0
    bes = [dm.HipBackend(pkg, p, rho[0], rho[1], device_index=0) for p in parts]
    bufs = []
    for p, be in zip(parts, bes):
        for name in ("v_curr", "rho"):
            be.upload(name, p.slice_global(st[name]))
        bufs.append([be.new_buffer(sum(pl[2])) for pl in (p.plan_send_v, p.plan_recv_v)])

    def exchange(send_slot, recv_slot, send_plan, recv_plan):
        # all_to_all_single semantics: block q of rank r's send buffer -> block r of rank q's recv buffer
        for r in range(world):
            so = np.concatenate([[0], np.cumsum(getattr(parts[r], send_plan)[2])])
            for q in range(world):
                ro = np.concatenate([[0], np.cumsum(getattr(parts[q], recv_plan)[2])])
                n = so[q + 1] - so[q]
                assert n == ro[r + 1] - ro[r]
                if n:
                    bufs[q][recv_slot][ro[r]:ro[r + 1]] = bufs[r][send_slot][so[q]:so[q + 1]]

    env = pkg.AdmmEnv("single", rho[0], rho[1], verbose=0, grid=g)
    single = pkg.ModelAcopf(env)
    ref = oracle.OracleModel(g, *rho)
    beta = 1e3
    for it in range(1, 7):
        for be, b in zip(bes, bufs):
            be.update_x(it)
            be.update_xbar()
            be.pack(0, b[0])
        torch.cuda.synchronize()
        exchange(0, 1, "plan_send_v", "plan_recv_v")
        tot = torch.zeros(4, dtype=torch.float64, device="cuda")
        for be, b in zip(bes, bufs):
            be.unpack(1, b[1])
            tot += be.update_zl_partial(beta)
        torch.cuda.synchronize()
        norms = torch.sqrt(tot).cpu().numpy()
        n1, _, _ = single.run_fixed(1, first_major_iter=it, beta=beta)
        ref.inner_prestep(); ref.update_x(it); ref.update_xbar(); ref.update_z(); ref.update_l()
        r = ref.update_residual()
        assert np.allclose(norms, n1, rtol=1e-10), (it, norms, n1)
        assert np.allclose(norms, r, rtol=1e-8), (it, norms, r)
    for name in ("u_curr", "v_curr", "z_curr", "l_curr"):
        glob = np.zeros(parts[0].nvar_global)
        for p, be in zip(parts, bes):
            glob[p.local_index_map()[:p.nvar_owned]] = be.download(name)[:p.nvar_owned]
        one = getattr(single.solution, name)
        assert np.abs(glob - one).max() <= 1e-10 * max(1.0, np.abs(one).max()), name


@pytest.mark.parametrize("seed,noise", [(0, 0.0), (1, 0.02), (2, 0.2)])
def test_branch_kernel_vs_oracle_on_perturbed_states(pkg, oracle, syn, seed, noise):
    """x-update alone on ~20k branch problems per case: realistic ADMM state (oracle, a few
    iterations) plus noise on multipliers / consensus targets / warm start, uploaded to the GPU;
    u within 1e-7 and the per-branch (auglag, TRON minor) iteration counts equal to the oracle's
    for >= 99.5% of the branches.  Exercises bound-active and many-auglag-iteration paths."""
    g = syn.sized_like("case13659pegase", seed=seed)
    rho = (5e1, 5e3)
    ref = oracle.OracleModel(g, *rho)
    for it in range(1, 4 + seed):
        ref.inner_prestep(); ref.update_x(it); ref.update_xbar(); ref.update_z(); ref.update_l(); ref.update_residual()
    rng = np.random.default_rng(100 + seed)
    G_, L_ = g["ngen"], g["nline"]
    if noise > 0:
        nb = 8 * L_
        ref.sol["l_curr"][2 * G_:] += noise * 50 * rng.normal(size=nb)
        ref.sol["v_curr"][2 * G_:] += noise * 0.2 * rng.normal(size=nb)
        ub = ref.sol["u_curr"][2 * G_:].reshape(L_, 8)
        ub[:, :4] += noise * 0.5 * rng.normal(size=(L_, 4))
        ub[:, 4:6] = np.clip(ub[:, 4:6] + noise * 0.1 * rng.normal(size=(L_, 2)), 0.7, 1.3)
        ref.membuf[:, 24:26] += noise * rng.normal(size=(L_, 2))
    env = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    for name in ("u_curr", "v_curr", "z_curr", "l_curr", "rho"):
        setattr(mod.solution, name, ref.sol[name])
    mod.membuf = ref.membuf.T
    major = 7
    mod.info.inner = major
    pkg.admm_update_x(env, mod)
    a_gpu, m_gpu, _ = mod.branch_counters()

    import ctypes as C
    L = oracle.lib()
    a_ref = np.zeros(L_, dtype=np.int64)
    m_ref = np.zeros(L_, dtype=np.int64)
    # per-branch oracle counters: run the oracle branch by branch through its single-branch entry
    u0 = ref.sol["u_curr"].copy()
    Y = np.stack([g[k] for k in ("YffR", "YffI", "YftR", "YftI", "YttR", "YttI", "YtfR", "YtfI")], 1).copy()
    s = ref.sol
    u_ref = u0.copy()
    for b in range(L_):
        p0 = 2 * G_ + 8 * b
        mb = ref.membuf[b]
        xl = np.array([g["FrVmBound"][2 * b], g["ToVmBound"][2 * b], g["FrVaBound"][2 * b], g["ToVaBound"][2 * b], -mb[28], -mb[28]])
        xu = np.array([g["FrVmBound"][2 * b + 1], g["ToVmBound"][2 * b + 1], g["FrVaBound"][2 * b + 1], g["ToVaBound"][2 * b + 1], 0.0, 0.0])
        ub_ = u0[p0:p0 + 8]
        x = np.array([np.sqrt(ub_[4]), np.sqrt(ub_[5]), ub_[6], ub_[7], -(ub_[0] ** 2 + ub_[1] ** 2), -(ub_[2] ** 2 + ub_[3] ** 2)])
        x = np.minimum(xu, np.maximum(xl, x))
        prm = np.zeros(31)
        prm[0:8] = s["l_curr"][p0:p0 + 8]
        prm[8:16] = s["rho"][p0:p0 + 8]
        prm[16:24] = s["v_curr"][p0:p0 + 8] - s["z_curr"][p0:p0 + 8]
        prm[24:27] = mb[24:27]
        it_, mi_ = C.c_int(), C.c_int()
        L.oracle_branch_solve(oracle._dp(x), oracle._dp(xl), oracle._dp(xu), oracle._dp(prm), oracle._dp(Y[b]), 1e-4, 50, 1e8,
                              C.cast(C.byref(it_), oracle.c_int_p), C.cast(C.byref(mi_), oracle.c_int_p))
        a_ref[b], m_ref[b] = it_.value, mi_.value
        cs, sn = np.cos(x[2] - x[3]), np.sin(x[2] - x[3])
        vv = x[0] * x[1]
        u_ref[p0:p0 + 8] = [Y[b, 0] * x[0] ** 2 + Y[b, 2] * vv * cs + Y[b, 3] * vv * sn,
                            -Y[b, 1] * x[0] ** 2 - Y[b, 3] * vv * cs + Y[b, 2] * vv * sn,
                            Y[b, 4] * x[1] ** 2 + Y[b, 6] * vv * cs - Y[b, 7] * vv * sn,
                            -Y[b, 5] * x[1] ** 2 - Y[b, 7] * vv * cs - Y[b, 6] * vv * sn,
                            x[0] ** 2, x[1] ** 2, x[2], x[3]]
    u_gpu = mod.solution.u_curr
    err = np.abs(u_gpu[2 * G_:] - u_ref[2 * G_:]).reshape(L_, 8).max(axis=1)
    same = (a_gpu == a_ref) & (m_gpu == m_ref)
    print("seed %d noise %g: %d of %d branches with other iteration counts than the oracle's; max auglag %d, max TRON minor %d"
          % (seed, noise, int((~same).sum()), L_, int(a_ref.max()), int(m_ref.max())))
    # the shipped (fast-arithmetic) build: measured 0 / 0 / 5 of 20 467 on these three states (tests/test_gpu_exact_arith.py reports them;
    # the exact-arithmetic build: 0) -- the gate is 10 of 20 467 = 5e-4, not the 0.5 % of rounds 1-5
    assert int((~same).sum()) <= 10, (int((~same).sum()), a_gpu.max(), a_ref.max())
    assert (err[same] <= 1e-7).all(), err[same].max()
    assert np.quantile(err, 0.999) <= 1e-5 and np.isfinite(u_gpu).all()
    assert a_ref.max() >= 3   # the sample does contain multi-iteration augmented-Lagrangian solves


def test_branch_kernel_on_the_first_iterations_of_outer_iteration_2__long_tron_paths_and_active_bounds(pkg, oracle, syn):
    """Where the solver works hardest: the first iterations of an outer iteration restart every branch's penalty at 10
    (acopf_auglag_linelimit_kernel_cpu.jl:74-79, major_iter == 1) and TRON makes 20 - 25 trips on the thermal-limit-binding branches
    (profiles/archive_r03_r04/r04_iteration_cost_vs_iteration_index_case13659_whole_solve.txt) -- the samples of the tests above (states after 3 - 6 iterations
    of outer iteration 1, plus noise) only had to contain an augmented-Lagrangian count of 3.  Here the GPU runs outer
    iteration 1 of the headline-size grid to its stop (step-wise control as admm_two_level.jl:29-77), updates lz and beta, and the x-updates of
    iterations 1, 2, 3 of OUTER ITERATION 2 are compared branch by branch with the oracle's single-branch solve from the same (downloaded)
    state: the sample must CONTAIN branches with >= 10 TRON minor iterations and branches that end on a bound (a binding thermal limit:
    s = -rateA; a voltage magnitude at its box), and those groups must AGREE with the oracle like the rest -- iteration counts equal on all but
    a handful, u within 1e-7 wherever they are."""
    from branch_ref import oracle_branch_subset
    g = syn.sized_like("case13659pegase", seed=0)
    rho = (5e1, 5e3)
    env = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    G_, L_ = int(g["ngen"]), int(g["nline"])
    beta = env.params.initial_beta
    # outer iteration 1 (admm_two_level.jl:29-77): norm of z, inner loop to its stop test, lz update, beta rule
    env.params.beta = beta
    mod.info.outer = 1
    pkg.admm_outer_prestep(env, mod)
    inner, res = mod.run_inner(1, 1000, beta)
    assert 50 <= inner < 1000, inner
    pkg.admm_update_lz(env, mod)
    if res[2] > env.params.theta * mod.info.norm_z_prev:
        beta = min(env.params.inc_c * beta, 1e24)
    env.params.beta = beta
    mod.info.outer = 2
    seen_long, seen_bound, total_mis = 0, 0, 0
    for it in (1, 2, 3):
        sol = {k: getattr(mod.solution, k).copy() for k in ("u_curr", "v_curr", "z_curr", "l_curr", "rho")}
        membuf = np.ascontiguousarray(mod.membuf.T).copy()
        u_ref, a_ref, m_ref, mult, x, xl, xu = oracle_branch_subset(oracle, g, sol, membuf, np.arange(L_), major_iter=it, want_x=True)
        mod.info.inner = it
        pkg.admm_inner_prestep(env, mod)
        pkg.admm_update_x(env, mod)
        a_gpu, m_gpu, _ = mod.branch_counters()
        u_gpu = mod.solution.u_curr[2 * G_:].reshape(L_, 8)
        err = np.abs(u_gpu - u_ref).max(axis=1)
        same = (a_gpu == a_ref) & (m_gpu == m_ref)
        long_path = m_ref >= 10
        on_bound = ((x == xl) | (x == xu))[:, [0, 1, 4, 5]].any(axis=1) & (xl != xu)[:, [0, 1, 4, 5]].any(axis=1)   # (x[4] = x[5] = 0 = xu is every
        on_bound &= ((x[:, 4] == xl[:, 4]) | (x[:, 5] == xl[:, 5]) | (x[:, 0] == xl[:, 0]) | (x[:, 0] == xu[:, 0]) |     #  idle branch: only the LOWER
                     (x[:, 1] == xl[:, 1]) | (x[:, 1] == xu[:, 1]))                                                  #  bound of s is a binding limit)
        print("outer 2 iteration %d: %d branches with other counts than the oracle's; %d with >= 10 TRON minor iterations (max %d, %d of them agree); "
              "%d ending on a bound (%d agree); max auglag %d" % (it, int((~same).sum()), int(long_path.sum()), int(m_ref.max()), int((same & long_path).sum()),
                                                                  int(on_bound.sum()), int((same & on_bound).sum()), int(a_ref.max())))
        assert int((~same).sum()) <= 10, (it, int((~same).sum()))
        assert (err[same] <= 1e-7).all(), (it, err[same].max())
        assert np.quantile(err, 0.999) <= 1e-5 and np.isfinite(u_gpu).all()
        for grp in (long_path, on_bound):
            if grp.sum():
                assert (same & grp).sum() >= grp.sum() - 5, (it, int(grp.sum()), int((same & grp).sum()))
        seen_long += int(long_path.sum())
        seen_bound += int(on_bound.sum())
        total_mis += int((~same).sum())
        # the rest of the iteration on the GPU (admm_two_level.jl:38-42), so that the next comparison starts from iteration it + 1's state
        pkg.admm_update_xbar(env, mod)
        pkg.admm_update_z(env, mod)
        pkg.admm_update_l(env, mod)
        pkg.admm_update_residual(env, mod)
    assert seen_long >= 100, seen_long     # the 20 - 25-trip TRON paths ARE in the sample ...
    assert seen_bound >= 20, seen_bound    # ... and so are solutions on a binding thermal limit / voltage bound
    mod.close()


def test_full_solve_to_convergence_matches_oracle(pkg, oracle, syn):
    """Whole two-level solve (case118-sized synthetic grid) to status Solved: same outer / cumulative
    iteration counts as the CPU path, objective within 1e-6 relative (north_star), u/v 1e-6 relative."""
    g = syn.sized_like("case118", seed=3)
    kw = dict(outer_iterlim=25, inner_iterlim=1000, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5)
    env, mod = pkg.solve_acopf("syn118", grid=g, verbose=0, **kw)
    ref = oracle.solve_acopf("syn118", grid=g, **kw)
    assert ref.info.status == 1 and mod.info.status == "Solved"
    assert (mod.info.outer, mod.info.cumul) == (ref.info.outer, ref.info.cumul)
    assert abs(mod.info.objval - ref.info.objval) <= 1e-6 * abs(ref.info.objval)
    for name in ("u_curr", "v_curr"):
        a, b = getattr(mod.solution, name), ref.sol[name]
        assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), name


def test_long_run_does_not_drift_from_oracle(pkg, oracle, syn):
    """3 outer x 1000 inner iterations on the case1354pegase-sized grid (lz updates and beta
    increases included): iteration counts equal, objective and ||Ax+By|| within 1e-6 relative
    (north_star).  The iterates themselves are only comparable up to the algorithm's own
    sensitivity: a branch solve is determined to gtol = 1e-6 and termination decisions amplify
    rounding over thousands of iterations.  That sensitivity is MEASURED here with the oracle
    alone (loads perturbed by 1e-13 relative move u by ~5e-5 after 2500 iterations); the GPU path
    (ocml vs glibc sincos rounding) must stay within 10x of it."""
    g = syn.sized_like("case1354pegase", seed=3)
    kw = dict(outer_iterlim=3, inner_iterlim=1000, rho_pq=1e1, rho_va=1e3)
    env, mod = pkg.solve_acopf("syn1354", grid=g, verbose=0, **kw)
    ref = oracle.solve_acopf("syn1354", grid=g, nthreads=8, **kw)
    g2 = dict(g)
    g2["Pd"] = g["Pd"] * (1 + 1e-13)
    ref2 = oracle.solve_acopf("syn1354", grid=g2, nthreads=8, **kw)
    assert (mod.info.outer, mod.info.cumul) == (ref.info.outer, ref.info.cumul)
    assert abs(mod.info.objval - ref.info.objval) <= 1e-6 * abs(ref.info.objval)
    assert np.isclose(mod.info.mismatch, ref.info.mismatch, rtol=1e-6)
    for name in ("u_curr", "v_curr", "l_curr"):
        a, b = getattr(mod.solution, name), ref.sol[name]
        inherent = np.abs(ref2.sol[name] - b).max()
        assert np.abs(a - b).max() <= 10 * inherent + 1e-9 * max(1.0, np.abs(b).max()), (name, inherent)
        assert np.abs(a - b).max() <= 1e-3 * max(1.0, np.abs(b).max()), name


def test_rolling_horizon_matches_oracle(pkg, oracle, case9_grid, tmp_path):
    """solve_acopf_rolling (row f3): 4 periods of case9 with a load profile read from .Pd/.Qd
    files, ramp-limited generator bounds and warm starts; per-period status, iteration counts
    and objective against the same procedure on the CPU oracle."""
    T = 4
    fac = np.array([1.0, 1.01, 0.995, 1.015])   # within the 2 %-of-Pmax ramp capability
    pd = np.outer(case9_grid["Pd"], fac)
    qd = np.outer(case9_grid["Qd"], fac)
    np.savetxt(str(tmp_path / "load.Pd"), pd)
    np.savetxt(str(tmp_path / "load.Qd"), qd)
    kw = dict(rho_pq=4e2, rho_va=4e4, outer_iterlim=25, outer_eps=2e-5, tight_factor=1.0)
    env, mod, res = pkg.solve_acopf_rolling(CASE9, load_prefix=str(tmp_path / "load"), start_period=1, end_period=T,
                                            ramp_ratio=0.02, **kw)
    # the same procedure on the oracle (acopf_admm_rolling_cpu.jl:16-48)
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in case9_grid.items()}
    ref = oracle.OracleModel(g, 4e2, 4e4, outer_iterlim=25, outer_eps=2e-5)
    ramp = 0.02 * g["pgmax"]
    for t in range(T):
        ref.g["Pd"][:] = pd[:, t]
        ref.g["Qd"][:] = qd[:, t]
        info = ref.admm_two_level()
        assert res[t]["status"] == ("Solved" if info.status == 1 else "IterationLimit")
        assert (res[t]["outer"], res[t]["cumul"]) == (info.outer, info.cumul), t
        assert abs(res[t]["objval"] - info.objval) <= 1e-6 * abs(info.objval)
        pg = ref.sol["u_curr"][0:2 * g["ngen"]:2]
        ref.pgmin_curr[:] = np.maximum(g["pgmin"], pg - ramp)
        ref.pgmax_curr[:] = np.minimum(g["pgmax"], pg + ramp)
    assert np.abs(mod.solution.u_curr - ref.sol["u_curr"]).max() <= 1e-6
    assert np.allclose(mod.pgmin_curr, ref.pgmin_curr, atol=1e-8) and np.allclose(mod.pgmax_curr, ref.pgmax_curr, atol=1e-8)


@pytest.mark.parametrize("case,rho,iters", [("case1354pegase", (1e1, 1e3), 40), ("case_ACTIVSg70k", (3e4, 3e5), 12)])
def test_pushed_bus_records_leave_every_bit_where_the_gathers_leave_it(pkg, syn, case, rho, iters):
    """The fused loop's bus launch reads what the branch launch pushed into its per-component records (k_bus_group_t<true>);
    EXAADMM_NO_REC=1 makes it gather from the solution planes like the step-wise path (k_bus_group_t<false>).  Same terms
    in the same order: u, v, z, l, the line-limit multipliers and the norms must be bit-identical -- also on the
    ACTIVSg70k-sized grid, whose hub buses have more than 8 components (the record slots of the later chunks)."""
    g = syn.sized_like(case, seed=0)
    outs = []
    for no_rec in (False, True):
        if no_rec:
            os.environ["EXAADMM_NO_REC"] = "1"
        try:
            env = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g)
            mod = pkg.ModelAcopf(env)
        finally:
            os.environ.pop("EXAADMM_NO_REC", None)
        norms, _, _ = mod.run_fixed(iters, first_major_iter=1, beta=1e3)
        outs.append((norms, mod.solution.u_curr, mod.solution.v_curr, mod.solution.z_curr, mod.solution.l_curr, mod.membuf))
        mod.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    assert np.isfinite(outs[0][0]).all() and outs[0][0][0] > 0


@pytest.mark.parametrize("case,rho,iters", [("case1354pegase", (1e1, 1e3), 30), ("case_ACTIVSg70k", (3e4, 3e5), 8)])
def test_lane_group_bus_kernels_equal_the_reference_order_twin_bit_for_bit(pkg, syn, case, rho, iters):
    """k_bus (one lane per bus, the reference's loops and summation order written out; EXAADMM_BUS_KERNEL=lane, which also selects the plain
    four-launch loop) against the default path (eight lanes per bus, the leader adding neutral-element-padded terms; records pushed by the
    branch launch; two-launch loop): u, v, z, l and the line-limit multipliers bit for bit -- also on the 88 207-branch grid whose hub buses
    take several chunks."""
    g = syn.sized_like(case, seed=0)
    outs = []
    for lane in (False, True):
        if lane:
            os.environ["EXAADMM_BUS_KERNEL"] = "lane"
        try:
            env = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g)
            mod = pkg.ModelAcopf(env)
        finally:
            os.environ.pop("EXAADMM_BUS_KERNEL", None)
        mod.run_fixed(iters, first_major_iter=1, beta=1e3)
        outs.append((mod.solution.u_curr, mod.solution.v_curr, mod.solution.z_curr, mod.solution.l_curr, mod.membuf))
        mod.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


def test_two_processes_sharing_the_gpu_reproduce_single_handle():
    """The N>1 code path with real processes: two ranks (torch.distributed.run, gloo with host-staged
    collectives, because RCCL refuses two ranks on one device) each drive their partition through
    HipBackend + DistributedAcopf.solve(); rank 0 compares with the single-handle solve."""
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "scripts", "dist_two_ranks_one_gpu.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    line = [l for l in out.splitlines() if l.startswith("two ranks on one GPU")]
    assert line and line[0].endswith("-> OK"), out[-2000:]


def _find_case(name):
    for d in (os.environ.get("MATPOWER_DATA", ""), GOLDEN):
        if d and os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return None


def test_case118_full_solve_anchor_gpu(pkg):
    """The reference's second anchor (acopf_update_cpu.jl:174-179) through the HIP path: 20 / 1232 / 129645.676."""
    a = G["solve_case118"]
    env, mod = pkg.solve_acopf(_find_case("case118.m"), outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5, verbose=0)
    assert mod.info.status == "Solved" and mod.info.outer == a["outer"] and mod.info.cumul == a["cumul"]
    assert abs(mod.info.objval - a["objval"]) <= 1e-6 * a["objval"]


def test_non_finite_input_does_not_stall_the_branch_kernel(pkg, syn):
    """A NaN in one branch's consensus target must cost that branch one solver tick, not the
    seconds its guarded loops would take to run out; every other branch is unaffected."""
    import time
    g = syn.sized_like("case1354pegase", seed=2)
    env = pkg.AdmmEnv("syn", 1e1, 1e3, verbose=0, grid=g)
    clean = pkg.ModelAcopf(env)
    dirty = pkg.ModelAcopf(env)
    for m in (clean, dirty):
        m.run_fixed(5, first_major_iter=1, beta=1e3)
    assert np.array_equal(clean.download("u_curr"), dirty.download("u_curr"))
    v = dirty.download("v_curr")
    b = 777
    v[2 * dirty.ngen + 8 * b + 1] = np.nan
    dirty.upload("v_curr", v)
    for m in (clean, dirty):
        m.info.inner = 6
    pkg.admm_update_x(env, clean)
    t0 = time.perf_counter()
    pkg.admm_update_x(env, dirty)
    dt = time.perf_counter() - t0
    assert dt < 0.05, dt                      # a healthy launch takes ~0.1 ms; the unguarded spin ~10 s
    uc, ud = clean.download("u_curr").reshape(-1), dirty.download("u_curr").reshape(-1)
    lo, hi = 2 * dirty.ngen + 8 * b, 2 * dirty.ngen + 8 * (b + 1)
    mask = np.ones(uc.size, dtype=bool)
    mask[lo:hi] = False
    assert np.array_equal(uc[mask], ud[mask])
    assert np.all(np.isfinite(ud[lo:hi]))      # the branch is left at its last accepted (finite) point
    a, m_, t = dirty.branch_counters()
    assert t[b] <= 2


def test_rolling_horizon_on_the_118_bus_system_matches_oracle(pkg, oracle, tmp_path):
    """solve_acopf_rolling (row f3) on real data: 3 periods of tests/golden/case118.m (first period = the reference's
    anchor solve, 20 / 1232), ramp-limited generator bounds and warm starts with the persistent line multipliers --
    per-period status, iteration counts and objective as the same procedure on the CPU oracle."""
    case118 = os.path.join(GOLDEN, "case118.m")
    g = oracle.grid_from_matpower(case118)
    T = 3
    fac = np.array([1.0, 1.004, 0.997])
    pd = np.outer(g["Pd"], fac)
    qd = np.outer(g["Qd"], fac)
    np.savetxt(str(tmp_path / "load.Pd"), pd)
    np.savetxt(str(tmp_path / "load.Qd"), qd)
    kw = dict(rho_pq=4e2, rho_va=4e4, outer_iterlim=25, outer_eps=2e-5, tight_factor=1.0)
    env, mod, res = pkg.solve_acopf_rolling(case118, load_prefix=str(tmp_path / "load"), start_period=1, end_period=T,
                                            ramp_ratio=0.02, **kw)
    assert (res[0]["status"], res[0]["outer"], res[0]["cumul"]) == ("Solved", 20, 1232)
    ref = oracle.OracleModel(g, 4e2, 4e4, outer_iterlim=25, outer_eps=2e-5)
    ramp = 0.02 * g["pgmax"]
    for t in range(T):
        ref.g["Pd"][:] = pd[:, t]
        ref.g["Qd"][:] = qd[:, t]
        info = ref.admm_two_level()
        assert res[t]["status"] == ("Solved" if info.status == 1 else "IterationLimit")
        assert (res[t]["outer"], res[t]["cumul"]) == (info.outer, info.cumul), (t, res[t], info.outer, info.cumul)
        assert abs(res[t]["objval"] - info.objval) <= 1e-6 * abs(info.objval)
        pg = ref.sol["u_curr"][0:2 * g["ngen"]:2]
        ref.pgmin_curr[:] = np.maximum(g["pgmin"], pg - ramp)
        ref.pgmax_curr[:] = np.minimum(g["pgmax"], pg + ramp)
    assert np.abs(mod.solution.u_curr - ref.sol["u_curr"]).max() <= 1e-6 * max(1.0, np.abs(ref.sol["u_curr"]).max())


@pytest.mark.parametrize("case,cumul,objval,tol", [("case9.m", 705, 5303.435, 1e-3), ("case118.m", 1232, 129645.676, 0.13)])
def test_plain_c_driver_solves_the_reference_anchors(case, cumul, objval, tol):
    """tools/exaadmm_solve (C, no Python in the loop): native MATPOWER loader + C ABI + fused two-level
    solve; prints the reference's statistics block.  Anchors (acopf_update_cpu.jl:168-179): Solved / 20 / 705 /
    5303.435 (atol 1e-3, the reference's) and Solved / 20 / 1232 / 129645.676 (rtol 1e-6)."""
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "exaadmm_solve")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "exaadmm.jl_amd", "csrc"), "../../tools/exaadmm_solve"])
    r = subprocess.run([exe, os.path.join(GOLDEN, case), "--outer", "25", "--outer_eps", "2e-5", "--quiet"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "Solved" in out
    assert int(re.search(r"Cumulative iterations[ .]*(\d+)", out).group(1)) == cumul
    assert int(re.search(r"Outer iterations[ .]*(\d+)", out).group(1)) == 20
    assert abs(float(re.search(r"Objective value[ .]*([-+0-9.eE]+)", out).group(1)) - objval) <= tol + 0.5e-6 * objval   # (the block prints 7 digits)
    assert abs(float(re.search(r"^objval=([-+0-9.eE]+)", out, flags=re.M).group(1)) - objval) <= tol


def test_relaxed_check_every_is_opt_in_and_bounded(pkg):
    """check_every = k > 1 (opt-in relaxation): the inner-loop test is evaluated every k-th iteration only,
    so each inner loop runs at most k-1 iterations longer than with the reference semantics."""
    env0, ref = pkg.solve_acopf(CASE9, outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5, verbose=0)
    env1, rel = pkg.solve_acopf(CASE9, outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5, verbose=0, check_every=4)
    assert ref.info.cumul == 705 and ref.info.status == "Solved"
    assert rel.info.status == "Solved"
    assert rel.info.cumul % 4 == 0 or rel.info.inner % 4 == 0
    assert 705 - 3 * 25 <= rel.info.cumul <= 705 + 3 * 25
    assert abs(rel.info.objval - ref.info.objval) <= 1e-4 * abs(ref.info.objval)   # a different (longer) trajectory to the same tolerance


def test_plain_c_driver_rolling_horizon_matches_python_mirror(pkg, case9_grid, tmp_path):
    """tools/exaadmm_solve --rolling: the rolling-horizon procedure (row f3) driven from C through
    exaadmm_set_load / exaadmm_set_pg_bounds gives the per-period iteration counts of the Python mirror."""
    import re
    import subprocess
    from conftest import ROOT
    fac = np.array([1.0, 1.01, 0.995, 1.015])
    np.savetxt(str(tmp_path / "load.Pd"), np.outer(case9_grid["Pd"], fac), fmt="%.17g")
    np.savetxt(str(tmp_path / "load.Qd"), np.outer(case9_grid["Qd"], fac), fmt="%.17g")
    env, mod, res = pkg.solve_acopf_rolling(CASE9, load_prefix=str(tmp_path / "load"), start_period=1, end_period=4,
                                            ramp_ratio=0.02, rho_pq=4e2, rho_va=4e4, outer_iterlim=25, outer_eps=2e-5, tight_factor=1.0)
    exe = os.path.join(ROOT, "tools", "exaadmm_solve")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "exaadmm.jl_amd", "csrc"), "../../tools/exaadmm_solve"])
    r = subprocess.run([exe, CASE9, "--outer", "25", "--outer_eps", "2e-5", "--quiet", "--load_prefix", str(tmp_path / "load"),
                        "--start_period", "1", "--end_period", "4", "--rolling"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    cum = [int(x) for x in re.findall(r"Cumulative iterations[ .]*(\d+)", r.stdout)]
    obj = [float(x) for x in re.findall(r"Objective value[ .]*([-+0-9.eE]+)", r.stdout)]
    assert cum == [res[t]["cumul"] for t in range(4)]
    assert all(abs(obj[t] - res[t]["objval"]) <= 1e-5 * abs(res[t]["objval"]) for t in range(4))


def test_checkpoint_and_resume_through_the_abi(pkg, syn):
    """The state of the path is the Solution vectors + membuf rows 25-27/29 (+ the ramp-limited
    generator bounds): downloading them, building a NEW device model and uploading them resumes the
    iteration bit for bit (what a Julia-side checkpoint of mod.solution / mod.membuf relies on)."""
    g = syn.sized_like("case1354pegase", seed=5)
    env = pkg.AdmmEnv("syn", 1e1, 1e3, verbose=0, grid=g)
    a = pkg.ModelAcopf(env)
    a.run_fixed(30, first_major_iter=1, beta=1e3)
    names = ("u_curr", "v_curr", "z_curr", "z_prev", "l_curr", "lz", "rho")
    snap = {n: a.download(n) for n in names}
    mb = a.membuf.copy()
    ra = a.run_fixed(10, first_major_iter=31, beta=1e3)[0]

    b = pkg.ModelAcopf(env)                      # fresh handle = "restart of the process"
    for n in names:
        b.upload(n, snap[n])
    b.membuf = mb
    rb = b.run_fixed(10, first_major_iter=31, beta=1e3)[0]
    assert np.array_equal(ra, rb)
    for n in names:
        assert np.array_equal(a.download(n), b.download(n)), n
    assert np.array_equal(a.membuf, b.membuf)


class _OneRank:
    """stand-in for torch.distributed with a single rank (no process group needed)"""

    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def get_rank():
        return 0

    @staticmethod
    def get_backend():
        return "nccl"


def test_native_distributed_loop_at_one_rank_matches_fused_solve(pkg, syn):
    """csrc/dist_native.h: the distributed inner loop inside the library (RCCL bound at run time; with one rank
    there is no communicator) must reproduce the fused single-GPU solve: same iteration counts, u / v to 1e-10."""
    dm = exaadmm_loader.load_submodule("distributed")
    g = syn.sized_like("case1354pegase", seed=2)
    rho = (1e1, 1e3)
    part = dm.Partition(g, 1, 0)
    st = dm.init_state(g, *rho)
    be = dm.HipBackend(pkg, part, rho[0], rho[1], device_index=0)
    par = dict(dm.DEFAULT_PAR)
    par.update(outer_iterlim=3, inner_iterlim=200)
    model = dm.DistributedAcopf(part, be, _OneRank, st, par)
    be.attach_native(_OneRank)
    info = model.solve()
    env, mod = pkg.solve_acopf("syn", grid=g, rho_pq=rho[0], rho_va=rho[1], outer_iterlim=3, inner_iterlim=200, verbose=0)
    assert (info["outer"], info["cumul"]) == (mod.info.outer, mod.info.cumul)
    assert abs(info["mismatch"] - mod.info.mismatch) <= 1e-10 * mod.info.mismatch
    for name in ("u_curr", "v_curr", "z_curr"):
        a, b = be.download(name), getattr(mod.solution, name)
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), name
    # the same loop with a REAL (one-rank) RCCL communicator attached, plus the self test that sends / receives through a
    # group on the handle's stream -- proves RCCL and the library share a runtime
    os.environ["EXAADMM_DIST_ONE_RANK_COMM"] = "1"
    try:
        be1 = dm.HipBackend(pkg, part, rho[0], rho[1], device_index=0)
        m1 = dm.DistributedAcopf(part, be1, _OneRank, st, par)
        be1.attach_native(_OneRank)
    finally:
        del os.environ["EXAADMM_DIST_ONE_RANK_COMM"]
    assert b"rccl" in pkg.library().exaadmm_dist_rccl_path()
    import ctypes as C
    err = C.c_double(1.0)
    pkg._check(pkg.library().exaadmm_dist_selftest(be1.mod.handle, 4096, C.byref(err)), "exaadmm_dist_selftest")
    assert err.value == 0.0
    # ... and the same ring captured into a hipGraph and replayed (what a partition's batches do with their exchanges): real RCCL
    # must be capturable in this runtime
    cap = C.c_int(-1)
    pkg._check(pkg.library().exaadmm_dist_capture_selftest(be1.mod.handle, 4096, C.byref(cap)), "exaadmm_dist_capture_selftest")
    print("real RCCL (one rank) captured into a hipGraph and replayed:", cap.value)
    assert cap.value == 1
    info1 = m1.solve()
    assert (info1["outer"], info1["cumul"]) == (info["outer"], info["cumul"]) and info1["mismatch"] == info["mismatch"]
    # fixed-count mode (benchmark path) advances the same trajectory as the single-GPU fixed-count loop
    be2 = dm.HipBackend(pkg, part, rho[0], rho[1], device_index=0)
    for name in ("v_curr", "rho"):
        be2.upload(name, part.slice_global(st[name]))
    be2.attach_native(_OneRank)
    res, ms = be2.run_fixed_native(12, 1, 1e3)
    env3 = pkg.AdmmEnv("syn", rho[0], rho[1], verbose=0, grid=g)
    m3 = pkg.ModelAcopf(env3)
    n3, _, _ = m3.run_fixed(12, first_major_iter=1, beta=1e3)
    assert np.allclose(res, n3, rtol=1e-10) and ms > 0
