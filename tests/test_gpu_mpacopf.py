"""GPU parity tests of the multi-period ACOPF model: the time-expanded HIP
model, called through the C ABI, against the reference's own known answers
(test/algorithms/mpacopf_update_cpu.jl via tests/golden/mpacopf_case9.json) and against the CPU
oracle (oracle/mpacopf_oracle.c) on the same inputs.  Tolerances next to each assertion."""
import json
import os

import numpy as np
import pytest

from conftest import CASE9, GOLDEN
import exaadmm_loader

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(GOLDEN, "mpacopf_case9.json")))


@pytest.fixture(scope="module")
def pkg():
    p = exaadmm_loader.load_package()
    p.library()
    return p


@pytest.fixture(scope="module")
def syn():
    return exaadmm_loader.load_submodule("synthetic")


def _case9_loads(grid):
    sc = G["load_scaling"]
    pd = np.stack([grid["Pd"] * a for a in sc["P"]], 1)
    qd = np.stack([grid["Qd"] * a for a in sc["Q"]], 1)
    return pd, qd


def _check(mod, name, key, atol):
    for t in range(3):
        full = np.concatenate([G[key + "_GEN"][t], G[key + "_BR"][t]])
        got = getattr(mod.models[t].solution, name)
        assert np.abs(got - full).max() <= atol, (key, t, np.abs(got - full).max())


def test_case9_three_periods_one_iteration_reference_known_answers(pkg):
    """Same sequence and vectors as test/algorithms/mpacopf_update_cpu.jl:1-427, atol 1e-6."""
    env = pkg.AdmmEnv(CASE9, 4e2, 4e4, verbose=0)
    mod = pkg.ModelMpacopf(env, _case9_loads(env.data), end_period=3)
    assert mod.nvar == 78 + 3 and mod.len_horizon == 3
    env.params.scale, env.params.initial_beta, env.params.beta = 1e-4, 1e3, 1e3
    pkg.admm_increment_outer(env, mod)
    pkg.admm_outer_prestep(env, mod)
    pkg.admm_increment_reset_inner(env, mod)
    pkg.admm_increment_inner(env, mod)
    pkg.admm_inner_prestep(env, mod)
    atol = G["atol"] + 1e-9   # the reference test's tolerance; its literals are truncated at 6 decimals
    pkg.admm_update_x(env, mod)
    _check(mod, "u_curr", "U", atol)
    pkg.admm_update_xbar(env, mod)
    _check(mod, "v_curr", "V", atol)
    pkg.admm_update_z(env, mod)
    _check(mod, "z_curr", "Z", atol)
    pkg.admm_update_l(env, mod)
    _check(mod, "l_curr", "L", atol + 1e-6)
    pkg.admm_update_residual(env, mod)
    G3 = mod.ngen
    for t in range(3):          # residual identities, mpacopf_update_cpu.jl:393-407
        s = mod.models[t].solution
        u, v, z, zp = s.u_curr, s.v_curr, s.z_curr, s.z_prev
        assert np.abs(s.rp - (u - v + z)).max() <= atol
        assert np.abs(s.rd - (z - zp)).max() <= atol
        assert np.abs(s.Ax_plus_By - (u - v)).max() <= atol
        if t > 0:
            r = mod.solution[t]
            vprev = mod.models[t - 1].solution.v_curr[0:2 * G3:2]
            assert np.abs(r.rp - (r.u_curr - vprev + r.z_curr)).max() <= atol
            assert np.abs(r.rd - (r.z_curr - r.z_prev)).max() <= atol
            assert np.abs(r.Ax_plus_By - (r.u_curr - vprev)).max() <= atol
    lz0 = [mod.models[t].solution.lz for t in range(3)]
    rlz0 = [mod.solution[t].lz for t in range(3)]
    pkg.admm_update_lz(env, mod)
    for t in range(3):          # lz recurrence, :409-427
        s = mod.models[t].solution
        assert np.abs(s.lz - (lz0[t] + env.params.beta * s.z_curr)).max() <= atol
        if t > 0:
            r = mod.solution[t]
            assert np.abs(r.lz - (rlz0[t] + env.params.beta * r.z_curr)).max() <= atol


@pytest.mark.parametrize("fused", [True, False])
def test_case9_three_periods_full_solve_anchor(pkg, fused):
    """mpacopf_update_cpu.jl:429-433: Solved, outer 20, cumul 729, objval 15901.48 (atol 1e-2)."""
    a = G["solve"]
    env = pkg.AdmmEnv(CASE9, 4e2, 4e4, verbose=0)
    env2, mod = pkg.solve_mpacopf(CASE9, load=_case9_loads(env.data), end_period=3, warm_start=False,
                                  outer_iterlim=a["outer_iterlim"], rho_pq=4e2, rho_va=4e4, outer_eps=a["outer_eps"],
                                  fused=fused)
    i = mod.info
    assert i.status == "Solved"
    assert i.outer == a["outer"]
    assert i.cumul == a["cumul"]
    assert abs(i.objval - a["objval"]) <= a["objval_atol"]


def _oracle_mp(oracle, grid, pd, qd, rho_pq, rho_va, **kw):
    og = {k: (np.asarray(v, dtype=np.int32) if isinstance(v, np.ndarray) and v.dtype == np.int64 else v) for k, v in grid.items()}
    return oracle.OracleMpModel(og, pd, qd, rho_pq, rho_va, **kw)


def test_multi_period_trajectory_matches_oracle(pkg, syn, oracle):
    """Synthetic 118-bus-sized grid, T = 4 periods with different load levels: 12 step-wise inner
    iterations, every vector (period blocks and ramp blocks) within 1e-8 of the CPU oracle."""
    grid = syn.sized_like("case118", seed=3)
    T = 4
    rng = np.random.default_rng(5)
    al = 1.0 + 0.02 * rng.standard_normal(T)
    pd = np.stack([grid["Pd"] * a for a in al], 1)
    qd = np.stack([grid["Qd"] * a for a in al], 1)
    env = pkg.AdmmEnv("syn118", 1e1, 1e3, verbose=0, grid=grid)
    mod = pkg.ModelMpacopf(env, (pd, qd), end_period=T, ramp_ratio=0.02)
    env.params.scale, env.params.beta = 1e-4, 1e3
    om = _oracle_mp(oracle, env.data, pd, qd, 1e1, 1e3, ramp_ratio=0.02, scale=1e-4)
    om.par.beta = 1e3
    Gn = mod.ngen
    mod.info.inner = 0
    for it in range(1, 13):
        pkg.admm_increment_inner(env, mod)
        pkg.admm_inner_prestep(env, mod); om.inner_prestep()
        pkg.admm_update_x(env, mod); om.update_x(it)
        pkg.admm_update_xbar(env, mod); om.update_xbar()
        pkg.admm_update_z(env, mod); om.update_z()
        pkg.admm_update_l(env, mod); om.update_l()
        pkg.admm_update_residual(env, mod); ores = om.update_residual()
        res = (mod.info.primres, mod.info.dualres, mod.info.norm_z_curr, mod.info.mismatch)
        assert np.allclose(res, ores, rtol=1e-9, atol=1e-12), (it, res, ores)
        if it in (1, 2, 6, 12):
            for t in range(T):
                for name in ("u_curr", "v_curr", "z_curr", "l_curr"):
                    got = getattr(mod.models[t].solution, name)
                    ref = om.periods[t].sol[name]
                    scale_ = max(1.0, np.abs(ref).max())
                    assert np.abs(got - ref).max() <= 1e-8 * scale_, (it, t, name, np.abs(got - ref).max())
                if t > 0:
                    for name in ("u_curr", "s_curr", "z_curr", "l_curr"):
                        got = getattr(mod.solution[t], name)
                        ref = om.ramp[t][name]
                        assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max()), (it, t, "ramp " + name)
                    assert np.abs(mod.solution[t].mu - om.gen_membuf[t][:, 6]).max() <= 1e-6 * max(1.0, np.abs(om.gen_membuf[t][:, 6]).max())
                    assert np.array_equal(mod.solution[t].xi, om.gen_membuf[t][:, 7])
    # outer-loop pieces
    pkg.admm_update_lz(env, mod); om.update_lz()
    pkg.admm_outer_prestep(env, mod)
    assert abs(mod.info.norm_z_prev - om.norm_z_prev()) <= 1e-9 * max(1.0, mod.info.norm_z_prev)
    pkg.admm_poststep(env, mod)
    assert abs(mod.info.objval - om.objective()) <= 1e-8 * abs(om.objective())


def test_multi_period_trajectory_at_1354_bus_size_matches_oracle(pkg, syn, oracle):
    """What `bench.py --periods 4` runs, at a size the serial oracle follows in seconds: the synthetic case1354pegase-sized
    grid (1 991 branches, 260 generators) over T = 4 load levels with the README's rho: 10 fused fixed-count iterations
    from the cold start, then every vector of every period and ramp block against the CPU oracle.  (Multi-period parity
    stopped at 118 buses before.)"""
    grid = syn.sized_like("case1354pegase", seed=0)
    T = 4
    al = 1.0 + 0.01 * np.sin(np.arange(T) * 2 * np.pi / T)       # the load profile of bench.py --periods
    pd = np.stack([grid["Pd"] * a for a in al], 1)
    qd = np.stack([grid["Qd"] * a for a in al], 1)
    env = pkg.AdmmEnv("syn1354", 1e1, 1e3, verbose=0, grid=grid)
    mod = pkg.ModelMpacopf(env, (pd, qd), end_period=T, ramp_ratio=0.02)
    env.params.scale, env.params.beta = 1e-4, 1e3
    om = _oracle_mp(oracle, env.data, pd, qd, 1e1, 1e3, ramp_ratio=0.02, scale=1e-4)
    om.par.beta = 1e3
    NIT = 10
    norms, _, _ = mod.run_fixed(NIT, first_major_iter=1, beta=1e3)
    for it in range(1, NIT + 1):
        om.inner_prestep(); om.update_x(it); om.update_xbar(); om.update_z(); om.update_l()
        ores = om.update_residual()
    assert np.allclose(norms, ores, rtol=1e-8, atol=1e-12), (norms, ores)
    worst = 0.0
    for t in range(T):
        for name in ("u_curr", "v_curr", "z_curr", "l_curr"):
            got = getattr(mod.models[t].solution, name)
            ref = om.periods[t].sol[name]
            err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, err)
            assert err <= 1e-7, (t, name, err)
        if t > 0:
            for name in ("u_curr", "s_curr", "z_curr", "l_curr"):
                got = getattr(mod.solution[t], name)
                ref = om.ramp[t][name]
                assert np.abs(got - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), (t, "ramp " + name)
            assert np.array_equal(mod.solution[t].xi, om.gen_membuf[t][:, 7])    # ramp penalties: exact
    print("1354-bus-sized grid x 4 periods, %d iterations: max relative deviation from the oracle %.2e" % (NIT, worst))


def test_multi_period_fused_loop_matches_oracle(pkg, syn, oracle):
    """Fused device-resident loop (hipGraph replay, device-side break test with the max-over-periods
    norm): same inner-iteration counts as the oracle's admm_two_level over 3 outer iterations."""
    grid = syn.sized_like("case118", seed=4)
    T = 3
    al = np.array([1.0, 1.01, 0.985])
    pd = np.stack([grid["Pd"] * a for a in al], 1)
    qd = np.stack([grid["Qd"] * a for a in al], 1)
    env = pkg.AdmmEnv("syn118", 1e1, 1e3, verbose=0, grid=grid)
    env2, mod = pkg.solve_mpacopf("syn118", load=(pd, qd), end_period=T, warm_start=False, outer_iterlim=3, inner_iterlim=400,
                                  rho_pq=1e1, rho_va=1e3, grid=grid)
    og = {k: (np.asarray(v, dtype=np.int32) if isinstance(v, np.ndarray) and v.dtype == np.int64 else v) for k, v in env.data.items()}
    om = oracle.solve_mpacopf(og, pd, qd, outer_iterlim=3, inner_iterlim=400, rho_pq=1e1, rho_va=1e3, warm_start=False)
    assert (mod.info.outer, mod.info.cumul) == (om.info.outer, om.info.cumul)
    assert abs(mod.info.objval - om.info.objval) <= 1e-6 * abs(om.info.objval)     # north_star tolerance
    assert abs(mod.info.mismatch - om.info.mismatch) <= 1e-6 * max(1.0, om.info.mismatch)
    for t in range(T):
        ref = om.periods[t].sol["u_curr"]
        assert np.abs(mod.models[t].solution.u_curr - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())


def test_multi_period_solve_on_the_118_bus_system_matches_oracle(pkg, oracle):
    """Real data: tests/golden/case118.m over 3 load levels with generator ramp coupling, cold start, fused loop:
    outer / cumulative iteration counts, objective and every period's u as the CPU oracle."""
    from conftest import CASE118
    grid = oracle.grid_from_matpower(CASE118)
    T = 3
    al = np.array([1.0, 1.006, 0.992])
    pd = np.stack([grid["Pd"] * a for a in al], 1)
    qd = np.stack([grid["Qd"] * a for a in al], 1)
    kw = dict(outer_iterlim=4, inner_iterlim=400, rho_pq=4e2, rho_va=4e4, warm_start=False)
    env2, mod = pkg.solve_mpacopf(CASE118, load=(pd, qd), end_period=T, **kw)
    om = oracle.solve_mpacopf(grid, pd, qd, **kw)
    assert (mod.info.outer, mod.info.cumul) == (om.info.outer, om.info.cumul)
    assert abs(mod.info.objval - om.info.objval) <= 1e-6 * abs(om.info.objval)
    for t in range(T):
        ref = om.periods[t].sol["u_curr"]
        assert np.abs(mod.models[t].solution.u_curr - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())


def test_single_period_multi_period_handle_equals_plain_handle(pkg, syn):
    """Edge: T = 1 has no ramp block and must reproduce the single-period model bit for bit."""
    grid = syn.sized_like("case118", seed=1)
    env = pkg.AdmmEnv("syn118", 1e1, 1e3, verbose=0, grid=grid)
    pd, qd = env.data["Pd"][:, None], env.data["Qd"][:, None]
    mp = pkg.ModelMpacopf(env, (pd, qd), end_period=1)
    sp = pkg.ModelAcopf(env)
    assert mp.nvar == sp.nvar
    r1 = mp.run_fixed(15, first_major_iter=1, beta=1e3)[0]
    r2 = sp.run_fixed(15, first_major_iter=1, beta=1e3)[0]
    assert np.array_equal(r1, r2)
    assert np.array_equal(mp.download("u_curr"), sp.download("u_curr"))


def test_warm_start_path_runs_and_converges_like_cold(pkg):
    """solve_mpacopf(warm_start=true): per-period solves first; only the line-limit multipliers
    survive the re-initialisation (reference behaviour).  On case9 no line limit binds, so the
    warm-started run must reproduce the cold anchor."""
    a = G["solve"]
    env = pkg.AdmmEnv(CASE9, 4e2, 4e4, verbose=0)
    env2, mod = pkg.solve_mpacopf(CASE9, load=_case9_loads(env.data), end_period=3, warm_start=True,
                                  outer_iterlim=a["outer_iterlim"], rho_pq=4e2, rho_va=4e4, outer_eps=a["outer_eps"])
    assert mod.info.status == "Solved"
    assert abs(mod.info.objval - a["objval"]) <= 1.0   # same optimum; iteration counts may differ by the carried multipliers


def test_ramp_vectors_round_trip_and_two_periods(pkg):
    """Edge: T = 2 (one ramp block) and upload / download of the SolutionRamping vectors."""
    env = pkg.AdmmEnv(CASE9, 4e2, 4e4, verbose=0)
    pd, qd = _case9_loads(env.data)
    mod = pkg.ModelMpacopf(env, (pd[:, :2], qd[:, :2]), end_period=2)
    assert mod.len_horizon == 2 and mod.nvar == 78 + 3
    rng = np.random.default_rng(0)
    for name in ("u_curr", "s_curr", "z_curr", "l_curr", "lz", "mu"):
        a = rng.standard_normal(2 * mod.ngen)
        mod.upload_ramp(name, a)
        assert np.array_equal(mod.download_ramp(name), a)
    # init_solution! resets the ramp vectors but not gen_membuf (multiplier / penalty), like the reference
    mu = mod.download_ramp("mu")
    mod.init_solution()
    assert np.array_equal(mod.download_ramp("mu"), mu)
    assert np.all(mod.download_ramp("z_curr") == 0.0)
    vprev = mod.models[0].solution.v_curr[0:2 * mod.ngen:2]
    assert np.array_equal(mod.solution[1].u_curr, vprev)
    assert np.array_equal(mod.solution[1].s_curr, -vprev)
    # a short two-period solve still converges to the sum of two nearly identical single-period optima
    env2, m2 = pkg.solve_mpacopf(CASE9, load=(pd[:, :2], qd[:, :2]), end_period=2, warm_start=False, outer_iterlim=25,
                                 rho_pq=4e2, rho_va=4e4, outer_eps=2e-5)
    assert m2.info.status == "Solved"
    assert abs(m2.info.objval - 2 * 5303.4) <= 0.01 * 2 * 5303.4


def test_plain_c_driver_solves_three_period_case9(tmp_path):
    """tools/exaadmm_solve --load_prefix: load files in the reference's format (nbus x T matrices),
    multi-period solve through the C ABI without Python; reference anchor 20 / 729 / 15901.48."""
    import re
    import subprocess
    from conftest import ROOT
    env_grid = exaadmm_loader.load_package().opf_loaddata(CASE9)
    pd, qd = _case9_loads(env_grid)
    np.savetxt(tmp_path / "load.Pd", pd, fmt="%.17g")
    np.savetxt(tmp_path / "load.Qd", qd, fmt="%.17g")
    exe = os.path.join(ROOT, "tools", "exaadmm_solve")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "exaadmm.jl_amd", "csrc"), "../../tools/exaadmm_solve"])
    r = subprocess.run([exe, CASE9, "--outer", "25", "--outer_eps", "2e-5", "--quiet", "--load_prefix", str(tmp_path / "load"),
                        "--start_period", "1", "--end_period", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "3 periods" in out and "Solved" in out
    assert int(re.search(r"Cumulative iterations[ .]*(\d+)", out).group(1)) == G["solve"]["cumul"]
    assert int(re.search(r"Outer iterations[ .]*(\d+)", out).group(1)) == G["solve"]["outer"]
    assert abs(float(re.search(r"Objective value[ .]*([-+0-9.eE]+)", out).group(1)) - G["solve"]["objval"]) <= 2e-2


def test_lane_per_bus_kernel_carries_the_ramp_terms_too(pkg, syn):
    """EXAADMM_BUS_KERNEL=lane (one lane per bus, the A/B twin of the lane-group kernel) must give
    bit-identical multi-period results: both evaluate the reference's ramp form of the generator terms."""
    grid = syn.sized_like("case118", seed=6)
    al = np.array([1.0, 0.99, 1.015])
    pd = np.stack([grid["Pd"] * a for a in al], 1)
    qd = np.stack([grid["Qd"] * a for a in al], 1)
    env = pkg.AdmmEnv("syn118", 1e1, 1e3, verbose=0, grid=grid)
    a = pkg.ModelMpacopf(env, (pd, qd), end_period=3)
    os.environ["EXAADMM_BUS_KERNEL"] = "lane"
    try:
        b = pkg.ModelMpacopf(env, (pd, qd), end_period=3)
    finally:
        del os.environ["EXAADMM_BUS_KERNEL"]
    ra = a.run_fixed(12, first_major_iter=1, beta=1e3)[0]
    rb = b.run_fixed(12, first_major_iter=1, beta=1e3)[0]
    assert np.array_equal(ra, rb)
    for name in ("u_curr", "v_curr", "z_curr"):
        assert np.array_equal(a.download(name), b.download(name)), name
    assert np.array_equal(a.download_ramp("u_curr"), b.download_ramp("u_curr"))


@pytest.mark.skipif(bool(os.environ.get("EXAADMM_MP_NO_SPEC") or os.environ.get("EXAADMM_NO_SPEC") or os.environ.get("EXAADMM_BUS_KERNEL") == "lane"),
                    reason="a switch of this run keeps the multi-period model on the four-launch loop")
@pytest.mark.parametrize("size,T", [("case118", 3), ("case1354pegase", 4)])
def test_fused_two_launch_loop_of_the_multi_period_model_equals_the_four_launch_loop(pkg, syn, size, T):
    """Round 4: the multi-period model runs the speculative two-launch loop (the z / l pass incl. the ramp block rides in the next
    branch launch's prologue, k_branch_t<true, true>; per-period residual sums reduced by the bus launch's extra block; the ramp
    subproblems' outputs rotate with the iterate) instead of four launches per iteration (EXAADMM_MP_NO_SPEC=1, A/B).  Every
    lane does the same arithmetic on the same inputs in both loops: u, v, z, l, lz of all periods and every ramp vector BITWISE
    equal after fixed-count runs of odd and even length and after convergence-tested loops; the norms (summed in a different
    order) to 1e-12; blocks that hold lanes of several periods are covered by the small case."""
    grid = syn.sized_like(size, seed=4)
    rng = np.random.default_rng(11)
    al = 1.0 + 0.02 * rng.standard_normal(T)
    load = (np.stack([grid["Pd"] * a for a in al], 1), np.stack([grid["Qd"] * a for a in al], 1))
    runs = []
    for four in (True, False):
        old = os.environ.get("EXAADMM_MP_NO_SPEC")
        if four:
            os.environ["EXAADMM_MP_NO_SPEC"] = "1"
        try:
            env = pkg.AdmmEnv("mp", 1e1, 1e3, verbose=0, grid=grid)
            mod = pkg.ModelMpacopf(env, load, end_period=T, ramp_ratio=0.02)
            assert mod.loop_kind() == (0 if four else 1)
            out, it = [], 1
            for K in (1, 2, 7, 24):
                norms, _, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
                it += K
                out.append([mod.download(n) for n in ("u_curr", "v_curr", "z_curr", "z_prev", "l_curr")] +
                           [mod.download_ramp(n) for n in ("u_curr", "s_curr", "z_curr", "z_prev", "l_curr", "mu", "xi")] + [mod.membuf.copy(), norms.copy()])
            for outer, limit in ((1, 21), (2, 150)):
                inner, res = mod.run_inner(outer, limit, 1e3)
                out.append([mod.download(n) for n in ("u_curr", "v_curr", "z_curr", "z_prev", "l_curr")] +
                           [mod.download_ramp(n) for n in ("u_curr", "s_curr", "z_curr", "z_prev", "l_curr", "mu", "xi")] + [mod.membuf.copy(), np.array([float(inner)]), res.copy()])
            runs.append(out)
            mod.close()
        finally:
            if old is None:
                os.environ.pop("EXAADMM_MP_NO_SPEC", None)
            else:
                os.environ["EXAADMM_MP_NO_SPEC"] = old
    for a, b in zip(runs[0], runs[1]):
        for k, (x, y) in enumerate(zip(a, b)):
            if k < 13 or (x.size == 1):
                assert np.array_equal(x, y), k
            else:     # the four norms: per-period sums added in another order
                assert np.allclose(x, y, rtol=1e-12, atol=0), (k, x, y)


def test_state_snapshot_of_a_multi_period_handle_restores_every_ramp_block(pkg, syn):
    """exaadmm_state_save / _restore on a multi-period handle (bench.py times repeated regions from ONE state): after a restore the
    same iterations give the same bits, period vectors and ramp blocks alike."""
    grid = syn.sized_like("case118", seed=6)
    T = 3
    load = (np.stack([grid["Pd"] * a for a in (1.0, 1.01, 0.99)], 1), np.stack([grid["Qd"] * a for a in (1.0, 1.01, 0.99)], 1))
    env = pkg.AdmmEnv("mp", 1e1, 1e3, verbose=0, grid=grid)
    mod = pkg.ModelMpacopf(env, load, end_period=T, ramp_ratio=0.02)
    mod.run_fixed(15, first_major_iter=1, beta=1e3)
    mod.state_save()

    def snap():
        return [mod.download(n) for n in ("u_curr", "v_curr", "z_curr", "z_prev", "l_curr")] + [mod.download_ramp(n) for n in ("u_curr", "s_curr", "z_curr", "z_prev", "l_curr", "mu", "xi")] + [mod.membuf.copy()]

    mod.run_fixed(9, first_major_iter=16, beta=1e3)
    a = snap()
    mod.run_fixed(20, first_major_iter=25, beta=1e3)
    mod.state_restore()
    mod.run_fixed(9, first_major_iter=16, beta=1e3)
    b = snap()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    mod.close()
