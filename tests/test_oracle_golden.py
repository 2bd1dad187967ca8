"""Pin the CPU oracle against the reference's own known answers
(/root/reference/test/algorithms/acopf_update_cpu.jl) and against outputs of the
reference's branch callbacks executed in the build container
(tests/golden/gen_branch_eval_vectors.py)."""
import json
import os

import numpy as np
import pytest

from conftest import CASE9, GOLDEN

G = json.load(open(os.path.join(GOLDEN, "case9_iter1.json")))
ATOL = G["atol"] + 1e-9  # 1e-6, the reference test's own tolerance (+ slack: literals are truncated)


def _one_iteration(oracle, grid):
    s = G["settings"]
    m = oracle.OracleModel(grid, s["rho_pq"], s["rho_va"], scale=s["scale"],
                           initial_beta=s["initial_beta"], beta=s["initial_beta"])
    m.info.outer = 1
    m.inner_prestep()
    return m


def test_case9_sizes(case9_grid):
    g = case9_grid
    assert (g["ngen"], g["nline"], g["nbus"]) == (3, 9, 9)
    assert g["bus_ref"] == 0
    # reference-bus angle pinned to (0,0) on branch 1's from side (opfdata.jl:704-713)
    assert tuple(g["FrVaBound"][:2]) == (0.0, 0.0)
    assert np.allclose(g["rateA"], [6.25, 6.25, 2.25, 9.0, 2.25, 6.25, 6.25, 6.25, 6.25])


def test_one_iteration_known_answers(oracle, case9_grid):
    m = _one_iteration(oracle, case9_grid)
    m.update_x(1)
    u = m.sol["u_curr"]
    assert np.abs(u[:6] - G["U_GEN"]).max() <= ATOL
    assert np.abs(u[6:] - G["U_BR"]).max() <= ATOL
    m.update_xbar()
    v = m.sol["v_curr"]
    assert np.abs(v[:6] - G["V_GEN"]).max() <= ATOL
    assert np.abs(v[6:] - G["V_BR"]).max() <= ATOL
    m.update_z()
    z = m.sol["z_curr"]
    assert np.abs(z[:6] - G["Z_GEN"]).max() <= ATOL
    assert np.abs(z[6:] - G["Z_BR"]).max() <= ATOL
    m.update_l()
    l = m.sol["l_curr"]
    # the reference literals are truncated (not rounded) at 6 decimals: allow 1e-6 + 1 ulp of print
    assert np.abs(l[:6] - G["L_GEN"]).max() <= ATOL + 1e-6
    assert np.abs(l[6:] - G["L_BR"]).max() <= ATOL + 1e-6
    # the w/theta multiplier entries are the only ones that depend on HOW tightly TRON
    # converged on the first solve; the reference test demands |l| <= 1e-6 there
    lw = l[6:].reshape(9, 8)[:, 4:]
    assert np.abs(lw).max() <= ATOL

    m.update_residual()
    s = m.sol
    assert np.abs(s["rp"] - (s["u_curr"] - s["v_curr"] + s["z_curr"])).max() <= ATOL
    assert np.abs(s["rd"] - (s["z_curr"] - s["z_prev"])).max() <= ATOL
    assert np.abs(s["Ax_plus_By"] - (s["u_curr"] - s["v_curr"])).max() <= ATOL
    lz_prev = s["lz"].copy()
    m.update_lz()
    assert np.abs(s["lz"] - (lz_prev + m.par.beta * s["z_curr"])).max() <= ATOL


def test_case9_full_solve_anchor(oracle):
    a = G["solve_case9"]
    m = oracle.solve_acopf(CASE9, outer_iterlim=a["outer_iterlim"], rho_pq=4e2, rho_va=4e4,
                           outer_eps=a["outer_eps"])
    assert m.info.status == 1            # :Solved
    assert m.info.outer == a["outer"]    # 20
    assert m.info.cumul == a["cumul"]    # 705
    assert abs(m.info.objval - a["objval"]) <= a["objval_atol"]


def test_case9_plumbing_config(oracle):
    """BASELINE.json configs[0]: outer=5 inner=50 CPU path."""
    m = oracle.solve_acopf(CASE9, outer_iterlim=5, inner_iterlim=50)
    assert m.info.outer == 5 and m.info.cumul <= 250 and np.isfinite(m.info.objval)


def test_branch_callbacks_match_reference_outputs(oracle):
    d = json.load(open(os.path.join(GOLDEN, "branch_eval_vectors.json")))
    L = oracle.lib()
    for c in d["cases"]:
        x, p, Y = (np.array(c[k]) for k in ("x", "param", "Y"))
        f = L.oracle_branch_eval_f(oracle._dp(x), oracle._dp(p), c["scale"], oracle._dp(Y))
        g = np.zeros(6)
        A = np.zeros(36)
        L.oracle_branch_eval_gh(oracle._dp(x), oracle._dp(p), c["scale"], oracle._dp(Y),
                                oracle._dp(g), oracle._dp(A))
        H, gr = np.array(c["H"]), np.array(c["g"])
        assert abs(f - c["f"]) <= 1e-12 * max(1.0, abs(c["f"]))
        assert np.abs(g - gr).max() <= 1e-12 * max(1.0, np.abs(gr).max())
        assert np.abs(A.reshape(6, 6) - H).max() <= 1e-12 * max(1.0, np.abs(H).max())
        assert np.array_equal(A.reshape(6, 6), A.reshape(6, 6).T)


def test_branch_gradient_finite_difference(oracle):
    d = json.load(open(os.path.join(GOLDEN, "branch_eval_vectors.json")))
    L = oracle.lib()
    c = d["cases"][0]
    x, p, Y = (np.array(c[k]) for k in ("x", "param", "Y"))
    g = np.zeros(6)
    A = np.zeros(36)
    L.oracle_branch_eval_gh(oracle._dp(x), oracle._dp(p), c["scale"], oracle._dp(Y),
                            oracle._dp(g), oracle._dp(A))
    h = 1e-6
    for i in range(6):
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        fp = L.oracle_branch_eval_f(oracle._dp(xp), oracle._dp(p), c["scale"], oracle._dp(Y))
        fm = L.oracle_branch_eval_f(oracle._dp(xm), oracle._dp(p), c["scale"], oracle._dp(Y))
        assert abs((fp - fm) / (2 * h) - g[i]) <= 1e-5 * max(1.0, abs(g[i]))


def test_openmp_oracle_matches_serial(oracle):
    a = oracle.solve_acopf(CASE9, outer_iterlim=3, inner_iterlim=30, nthreads=1)
    b = oracle.solve_acopf(CASE9, outer_iterlim=3, inner_iterlim=30, nthreads=4)
    assert a.info.cumul == b.info.cumul
    assert np.array_equal(a.sol["u_curr"], b.sol["u_curr"])


def _find_case(name):
    for d in (os.environ.get("MATPOWER_DATA", ""), GOLDEN, os.path.join(os.path.dirname(GOLDEN), "..", "data")):
        if d and os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return None


def test_case118_fixture_is_the_ieee_118_bus_system():
    """Integrity of tests/golden/case118.m: the published totals of the IEEE 118-bus system."""
    from oracle import oracle as O
    raw = O.read_matpower(_find_case("case118.m"))
    bus, gen, br, cost = (np.array(raw[k]) for k in ("mpc.bus", "mpc.gen", "mpc.branch", "mpc.gencost"))
    assert bus.shape[0] == 118 and gen.shape[0] == 54 and br.shape[0] == 186 and cost.shape[0] == 54
    assert bus[:, 2].sum() == 4242.0 and bus[:, 3].sum() == 1438.0          # MW, MVAr
    assert abs(gen[:, 8].sum() - 9966.2) < 1e-9                                # installed capacity, MW
    assert np.count_nonzero(bus[:, 5]) == 14                                   # bus shunts
    assert np.count_nonzero(br[:, 8]) == 9                                     # transformers (off-nominal ratio)
    assert int(bus[bus[:, 1] == 3, 0][0]) == 69                                # reference bus
    assert np.array_equal(np.unique(bus[:, 0]), np.arange(1, 119))
    zero = gen[:, 1] == 0
    assert np.all(cost[zero, 4] == 0.01) and np.all(cost[zero, 5] == 40.0)
    np.testing.assert_allclose(cost[~zero, 4], 10.0 / gen[~zero, 1], rtol=5e-6)
    assert np.all(cost[~zero, 5] == 20.0)


def test_case118_full_solve_anchor(oracle):
    """Second anchor of the reference test (acopf_update_cpu.jl:174-179): 20 outer / 1232 inner iterations,
    objective 129645.676 (rtol 1e-6) -- 186 branch subproblems through 1232 iterations with the same counts as the
    reference pins the aug-Lagrangian / TRON dynamics far harder than the 9-bus trajectories do."""
    a = G["solve_case118"]
    m = oracle.solve_acopf(_find_case("case118.m"), outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5)
    assert m.info.status == 1 and m.info.outer == a["outer"] and m.info.cumul == a["cumul"]
    assert abs(m.info.objval - a["objval"]) <= 1e-6 * a["objval"]
