"""-m gpu: the resident inner loop (kernels.h k_resident: ONE launch per batch of iterations, every wavefront resident, a bus
updated by the last lane to arrive at it, residual sums and the exact per-iteration stop test made by a control wavefront)
against the two-launch loop (EXAADMM_NO_RESIDENT=1).  Every lane does the arithmetic it does in the two-launch loop and every
sum runs in the same order, so everything must be BITWISE equal: u, v, z, l, multipliers / penalties, per-branch counters, the
four norms, the iteration at which the convergence-tested loop stops, whole solves.  Grids with more wavefronts than the
chip has SIMDs must fall back to the two-launch loop by themselves."""
import os

import numpy as np
import pytest

import exaadmm_loader

pytestmark = pytest.mark.gpu
SWITCHED_OFF = bool(os.environ.get("EXAADMM_NO_SPEC") or os.environ.get("EXAADMM_BUS_KERNEL") == "lane" or os.environ.get("EXAADMM_NO_REC")
                    or os.environ.get("EXAADMM_NO_RESIDENT") or os.environ.get("EXAADMM_RESIDENT") == "0")


@pytest.fixture(scope="module")
def pkg():
    p = exaadmm_loader.load_package()
    p.library()
    return p


class _two_launch:
    """environment switch for the duration of a block (the library reads it when a loop starts)"""
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("EXAADMM_NO_RESIDENT")
        if self.on:
            os.environ["EXAADMM_NO_RESIDENT"] = "1"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("EXAADMM_NO_RESIDENT", None)
        else:
            os.environ["EXAADMM_NO_RESIDENT"] = self.old


def _state(mod):
    s = mod.solution
    a, m, t = mod.branch_counters()
    return [s.u_curr.copy(), s.v_curr.copy(), s.z_curr.copy(), s.l_curr.copy(), np.ascontiguousarray(mod.membuf).copy(), a, m, t]


@pytest.mark.skipif(SWITCHED_OFF, reason="a switch of this run turns the resident loop off")
@pytest.mark.parametrize("case,rho", [("case1354pegase", (1e1, 1e3)), ("case13659pegase", (5e1, 5e3))])
def test_resident_loop_is_bitwise_the_two_launch_loop(pkg, case, rho):
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like(case, seed=0)
    runs = []
    for two in (True, False):
        with _two_launch(two):
            env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
            mod = pkg.ModelAcopf(env)
            assert mod.loop_kind() == (1 if two else 2)
            out, it = [], 1
            for K in (1, 2, 13, 60, 33):            # odd and even lengths, odd and even starting iterations
                norms, _, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
                it += K
                out.append(_state(mod) + [norms.copy()])
            for outer, limit in ((1, 45), (2, 7), (3, 200)):   # convergence-tested loop: stops by the limit and by the test, several batches
                inner, res = mod.run_inner(outer, limit, 1e3)
                out.append(_state(mod) + [res.copy(), np.array([inner])])
            runs.append(out)
            mod.close()
    for a, b in zip(runs[0], runs[1]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


@pytest.mark.skipif(SWITCHED_OFF, reason="a switch of this run turns the resident loop off")
def test_resident_whole_solves_reproduce_the_anchors_and_the_two_launch_runs(pkg):
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for case, outer, cumul in (("case9.m", 20, 705), ("case118.m", 20, 1232)):
        infos = []
        for two in (True, False):
            with _two_launch(two):
                env, mod = pkg.solve_acopf(os.path.join(golden, case), outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5, verbose=0)
                i = mod.info
                infos.append((i.status, i.outer, i.cumul, i.objval, i.primres, i.dualres, i.mismatch, mod.loop_kind(),
                              mod.solution.u_curr.copy(), mod.solution.v_curr.copy(), mod.solution.l_curr.copy()))
                mod.close()
        assert infos[0][:3] == ("Solved", outer, cumul) == infos[1][:3]
        assert (infos[0][7], infos[1][7]) == (1, 2)
        assert infos[0][3:7] == infos[1][3:7]
        for k in (8, 9, 10):
            assert np.array_equal(infos[0][k], infos[1][k])
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like("case13659pegase", seed=0)
    res = []
    for two in (True, False):
        with _two_launch(two):
            env, mod = pkg.solve_acopf("case13659pegase", grid=g, rho_pq=5e1, rho_va=5e3, outer_iterlim=20, inner_iterlim=1000, verbose=0)
            i = mod.info
            res.append((i.status, i.outer, i.cumul, i.objval, i.primres, mod.solution.u_curr.tobytes(), mod.solution.v_curr.tobytes()))
            mod.close()
    assert res[0] == res[1] and res[0][:3] == ("Solved", 3, 250)


def test_grids_with_more_wavefronts_than_simds_keep_the_two_launch_loop(pkg):
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like("case_ACTIVSg70k", seed=0)
    env = pkg.AdmmEnv("case_ACTIVSg70k", 3e4, 3e5, verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    assert mod.loop_kind() in (0, 1)       # 1 379 + 163 + 1 wavefronts do not fit 1 024 SIMDs
    mod.close()
