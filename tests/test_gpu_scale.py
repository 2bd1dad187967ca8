"""-m gpu tests at the two largest BASELINE.json configurations: ACTIVSg70k-sized (G 10 390 / L 88 207 /
B 70 000, rho 3e4 / 3e5, scale 1e-5) and the synthetic 1M-branch / 500k-bus grid.  At these sizes the
oracle cannot run whole iterations in test time, so the checks are (i) size-independent properties of the
device state and (ii) the branch kernel against the oracle's single-branch solve
(acopf_auglag_linelimit_kernel_cpu.jl:21-168) on a seeded sample of 20 000 branches, from the state the
GPU itself reached after three iterations.  What they guard: 32-bit index arithmetic, launches of
15 625 wavefronts, bus chunks of more than 8 incident ends, the bus-sorted permutation at scale."""
import numpy as np
import pytest

import exaadmm_loader
from branch_ref import oracle_branch_subset

pytestmark = pytest.mark.gpu

CONFIGS = {
    "case_ACTIVSg70k": dict(rho=(3e4, 3e5), scale=1e-5, grid=dict(cost="market")),
    "synthetic1M": dict(rho=(5e1, 5e3), scale=1e-4, grid=dict()),
}
_CACHE = {}


@pytest.fixture(scope="module")
def pkg():
    p = exaadmm_loader.load_package()
    p.library()
    return p


def _grid(case):
    if case not in _CACHE:
        syn = exaadmm_loader.load_submodule("synthetic")
        _CACHE.clear()    # one large grid at a time in host memory
        _CACHE[case] = syn.sized_like(case, seed=0, **CONFIGS[case]["grid"])
    return _CACHE[case]


def _model(pkg, case):
    cfg = CONFIGS[case]
    g = _grid(case)
    env = pkg.AdmmEnv(case, cfg["rho"][0], cfg["rho"][1], verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    env.params.scale = cfg["scale"]
    mod.sync_params()
    return g, env, mod


@pytest.mark.parametrize("case", list(CONFIGS))
def test_properties_at_scale(pkg, case):
    """bitwise determinism of the fused loop, residual identities, bus consensus, flow consistency, bounds"""
    outs = []
    for _ in range(2):
        g, env, mod = _model(pkg, case)
        norms, _, _ = mod.run_fixed(3, first_major_iter=1, beta=1e3)
        outs.append((mod.solution.u_curr, mod.solution.v_curr, norms.copy()))
        if len(outs) == 1:
            mod.close()
    (u, v, n0), (u1, v1, n1) = outs
    assert np.array_equal(u, u1) and np.array_equal(v, v1) and np.array_equal(n0, n1)   # bitwise
    assert np.isfinite(u).all() and np.isfinite(v).all()
    sol = mod.solution
    z, zp = sol.z_curr, sol.z_prev
    assert np.allclose(sol.rp, u - v + z, atol=1e-12)
    assert np.allclose(sol.rd, z - zp, atol=1e-12)
    nrm = np.sqrt([np.sum((u - v + z) ** 2), np.sum((z - zp) ** 2), np.sum(z ** 2), np.sum((u - v) ** 2)])
    assert np.allclose(n0, nrm, rtol=1e-9)    # reduction order differs from numpy's pairwise sum: 1e-9 relative
    G_, L_, B_ = int(g["ngen"]), int(g["nline"]), int(g["nbus"])
    vb = v[2 * G_:].reshape(L_, 8)
    bb = np.asarray(g["brBusIdx"]).reshape(L_, 2)
    ends_bus = np.concatenate([bb[:, 0], bb[:, 1]])
    assert np.bincount(ends_bus, minlength=B_).max() > 8      # the grid does exercise multi-chunk buses
    for col_fr, col_to in ((4, 5), (6, 7)):                    # w and theta: every end at a bus sees the same value
        ends_val = np.concatenate([vb[:, col_fr], vb[:, col_to]])
        at_bus = np.full(B_, np.nan)
        at_bus[ends_bus] = ends_val
        assert np.array_equal(at_bus[ends_bus], ends_val)
    # power balance of the bus update (acopf_bus_kernel_cpu.jl): sum pg - Pd/M = sum p + YshR w at every bus
    M = float(g["baseMVA"])
    gb = np.asarray(g["gen_bus"])
    pbal = np.bincount(gb, weights=v[0:2 * G_:2], minlength=B_) - np.asarray(g["Pd"]) / M
    pbal -= np.bincount(bb[:, 0], weights=vb[:, 0], minlength=B_) + np.bincount(bb[:, 1], weights=vb[:, 2], minlength=B_)
    wbus = np.zeros(B_)
    wbus[bb[:, 0]] = vb[:, 4]
    wbus[bb[:, 1]] = vb[:, 5]
    pbal -= np.asarray(g["YshR"]) * wbus
    assert np.abs(pbal).max() <= 1e-8 * max(1.0, np.abs(v).max())
    ub = u[2 * G_:].reshape(L_, 8)
    vi, vj, th = np.sqrt(ub[:, 4]), np.sqrt(ub[:, 5]), ub[:, 6] - ub[:, 7]
    pij = g["YffR"] * vi ** 2 + g["YftR"] * vi * vj * np.cos(th) + g["YftI"] * vi * vj * np.sin(th)
    assert np.allclose(ub[:, 0], pij, rtol=1e-9, atol=1e-9)
    assert (ub[:, 4:6] >= 0.81 - 1e-12).all() and (ub[:, 4:6] <= 1.21 + 1e-12).all()
    pg = u[0:2 * G_:2]
    assert (pg >= g["pgmin"] - 1e-15).all() and (pg <= g["pgmax"] + 1e-15).all()
    mod.close()


def _sample_against_oracle(pkg, oracle, case):
    """20 000 seeded branches: the GPU x-update of iteration 4 against the oracle's single-branch solve from the same state (the GPU's own
    state after 3 iterations, downloaded).  Returns the comparison (used with the shipped library in this process and with the
    reference-arithmetic build in a worker process)."""
    g, env, mod = _model(pkg, case)
    cfg = CONFIGS[case]
    mod.run_fixed(3, first_major_iter=1, beta=1e3)
    sol = {k: getattr(mod.solution, k) for k in ("u_curr", "v_curr", "z_curr", "l_curr", "rho")}
    membuf = np.ascontiguousarray(mod.membuf.T)
    G_, L_ = int(g["ngen"]), int(g["nline"])
    rng = np.random.default_rng(2024)
    # the sample always contains the first and last wavefronts of the launch and of the storage order
    sample = np.unique(np.concatenate([rng.choice(L_, 20000 - 256, replace=False), np.arange(128), np.arange(L_ - 128, L_)]))
    u_ref, a_ref, m_ref, mult = oracle_branch_subset(oracle, g, sol, membuf, sample, scale=cfg["scale"], major_iter=4)
    mod.info.inner = 4
    pkg.admm_update_x(env, mod)
    a_gpu, m_gpu, _ = mod.branch_counters()
    u_gpu = mod.solution.u_curr[2 * G_:].reshape(L_, 8)[sample]
    mb_gpu = mod.membuf.T[sample]
    err = np.abs(u_gpu - u_ref).max(axis=1)
    same = (a_gpu[sample] == a_ref) & (m_gpu[sample] == m_ref)
    mod.close()
    return dict(n=len(sample), same=same, err=err, u_gpu=u_gpu, mb_gpu=mb_gpu, mult=mult, a_ref=a_ref, m_ref=m_ref)


# count mismatches the shipped (fast-arithmetic) build may show on the 20 000-branch sample: measured 42 at ACTIVSg70k's settings (scale 1e-5 puts
# the projected-gradient test gtol = 1e-6 within rounding distance of more branches' final gradients) and 0 at 1M branches; the headline grid
# measures 0 - 5 of 20 467 per state (tests/test_gpu_exact_arith.py).  The reference-arithmetic build must show (essentially) none on the same
# samples -- which is what says the shipped build's mismatches are its arithmetic, not an error (test below).
FAST_BUILD_GATE = {"case_ACTIVSg70k": 64, "synthetic1M": 10}


@pytest.mark.parametrize("case", list(CONFIGS))
def test_branch_kernel_vs_oracle_on_a_sample_at_scale(pkg, oracle, case):
    """u within 1e-7 wherever the per-branch (augmented-Lagrangian, TRON minor) iteration counts agree -- on all but a measured handful of the
    20 000 (FAST_BUILD_GATE; rounds 1-5 allowed 0.5 %)."""
    r = _sample_against_oracle(pkg, oracle, case)
    same, err, u_gpu, mb_gpu, mult, a_ref, m_ref = (r[k] for k in ("same", "err", "u_gpu", "mb_gpu", "mult", "a_ref", "m_ref"))
    assert np.isfinite(u_gpu).all()
    print("%s: %d of %d sampled branches with other iteration counts than the oracle's (max auglag %d, max TRON minor %d)"
          % (case, int((~same).sum()), r["n"], int(a_ref.max()), int(m_ref.max())))
    assert int((~same).sum()) <= FAST_BUILD_GATE[case], (same.mean(), int((~same).sum()))
    assert (err[same] <= 1e-7).all(), err[same].max()
    assert np.quantile(err, 0.999) <= 1e-5
    # penalties are powers of ten: equal exactly wherever the counts agree.  The multipliers accumulate
    # mu * (constraint violation): a difference of d in the flows moves them by ~mu * d, so their tolerance is
    # the u tolerance (1e-7) times the branch's penalty, plus 1e-6 relative
    assert np.array_equal(mb_gpu[same, 26], mult[same, 2])
    dmult = np.abs(mb_gpu[same, 24:26] - mult[same, 0:2])
    assert (dmult <= 1e-6 * np.abs(mult[same, 0:2]) + 4e-7 * mult[same, 2:3]).all(), dmult.max()


def test_reference_arithmetic_build_on_the_70k_sample():
    """The same 20 000-branch sample of the ACTIVSg70k-sized grid through libexaadmm_hip_exact.so (-DEXA_EXACT_ORACLE_ARITH: every expression in
    the CPU path's operation order; a worker process, the package binds one library per process): at most 2 branches may take other iteration
    counts than the oracle (device libm), u within 1e-10 elsewhere."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    lib = os.path.join(ROOT, "exaadmm.jl_amd", "libexaadmm_hip_exact.so")
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, exaadmm_loader, test_gpu_scale as T\n"
            "from oracle import oracle as O\n"
            "pkg = exaadmm_loader.load_package(); pkg.library(); O.lib()\n"
            "assert '-DEXA_EXACT_ORACLE_ARITH' in pkg.library().exaadmm_version().decode()\n"
            "r = T._sample_against_oracle(pkg, O, 'case_ACTIVSg70k')\n"
            "print('RESULT ' + json.dumps({'n': r['n'], 'mismatches': int((~r['same']).sum()), 'max_err_same': float(r['err'][r['same']].max())}))\n"
            % (ROOT, os.path.join(ROOT, "tests")))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, EXAADMM_HIP_LIB=lib), capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    rep = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print("reference-arithmetic build, ACTIVSg70k-sized sample:", rep)
    assert rep["mismatches"] <= 2 and rep["max_err_same"] <= 1e-10, rep
