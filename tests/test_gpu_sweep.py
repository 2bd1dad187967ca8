"""-m gpu: the fused two-level loop (exaadmm_solve_two_level) against the CPU oracle on 14 seeded grids of three
sizes -- the robustness sweep that used to live in tests/tools/sweep_parity.py, now part of the suite.  Bar: the
SAME outer / cumulative iteration counts as the CPU path (every inner-loop break at the same iteration) and u
within 1e-6 relative (north_star's tolerance; observed <= 2e-7)."""
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
    for name in ("u_curr", "v_curr"):
        a, b = getattr(mod.solution, name), ref.sol[name]
        assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), name
    assert abs(mod.info.primres - ref.info.primres) <= 1e-7 * max(1.0, ref.info.primres)
    assert abs(mod.info.objval - ref.objective()) <= 1e-6 * max(1.0, abs(ref.objective()))
    mod.close()
