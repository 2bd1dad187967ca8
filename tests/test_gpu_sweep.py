"""-m gpu: the fused two-level loop (exaadmm_solve_two_level) against the CPU oracle on 14 seeded grids of three
sizes -- the robustness sweep that used to live in tests/tools/sweep_parity.py, now part of the suite.  Bar: the
SAME outer / cumulative iteration counts as the CPU path (every inner-loop break at the same iteration) and u / v
within 1e-6 relative on 99.9 % of the entries, 1e-5 on all (see the comment at the assertion)."""
import numpy as np
import pytest

import exaadmm_loader

pytestmark = pytest.mark.gpu

GRIDS = ([("case118", (4e2, 4e4), 150, s) for s in range(6)] +
         [("case1354pegase", (1e1, 1e3), 120, s) for s in range(6)] +
         [("case13659pegase", (5e1, 5e3), 60, s) for s in range(2)])


@pytest.fixture(scope="module")
def pkg():
    p = exaadmm_loader.load_package()
    p.library()
    return p


@pytest.mark.parametrize("case,rho,inner,seed", GRIDS, ids=["%s-seed%d" % (g[0], g[3]) for g in GRIDS])
def test_fused_solve_matches_oracle(pkg, oracle, case, rho, inner, seed):
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like(case, seed=seed)
    kw = dict(outer_iterlim=3, inner_iterlim=inner, rho_pq=rho[0], rho_va=rho[1], scale=1e-4)
    env, mod = pkg.solve_acopf(case, grid=g, verbose=0, **kw)
    ref = oracle.solve_acopf(case, grid=g, nthreads=8, **kw)
    assert (mod.info.outer, mod.info.cumul) == (ref.info.outer, ref.info.cumul)
    # These solves are CUT at `inner` iterations per outer iteration, far from converged (primres ~ 10).  Two floating-point
    # realisations of the algorithm then agree to ~1e-9 almost everywhere, but an isolated branch can differ by the
    # augmented-Lagrangian loop's own tolerance: a ratio test that flips on the last bit makes a later `cnorm <= eta` test
    # raise the penalty in one realisation and not in the other, and both solutions satisfy the 1e-6 constraint
    # tolerance (acopf_auglag_linelimit_kernel_cpu.jl:118-134).  Observed: max 1.06e-6 relative on one generator /
    # branch pair of the 13 659-bus grid, with the default AND with the exact-arithmetic build (which differs from the
    # oracle only by the device's sin / cos) -- so the bar is 1e-6 on the 99.9 % quantile and 1e-5 on the maximum.
    # Converged solves agree to 1e-9 (test_full_solve_to_convergence_matches_oracle, bench.py full_solve).
    for name in ("u_curr", "v_curr"):
        a, b = getattr(mod.solution, name), ref.sol[name]
        d = np.abs(a - b) / max(1.0, np.abs(b).max())
        assert np.quantile(d, 0.999) <= 1e-6 and d.max() <= 1e-5, (name, d.max())
    assert abs(mod.info.primres - ref.info.primres) <= 1e-7 * max(1.0, ref.info.primres)
    assert abs(mod.info.objval - ref.objective()) <= 1e-6 * max(1.0, abs(ref.objective()))
    mod.close()
