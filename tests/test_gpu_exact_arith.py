"""-m gpu: the reference-arithmetic twin of the library (libexaadmm_hip_exact.so, -DEXA_EXACT_ORACLE_ARITH)
against the oracle, and the default library against both.

On the host the exact build is bitwise the oracle (tests/test_device_code_on_host.py).  On the GPU the only
remaining difference is the device math library (sin / cos of the angle difference), so its per-branch
iteration counts must equal the oracle's on (essentially) every branch; the default build's distance from
it is then a measured number, not a tolerance.  Each library is driven in its own process (the package
binds one library per process through EXAADMM_HIP_LIB)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import exaadmm_loader
from oracle import oracle
from branch_ref import oracle_branch_subset
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
out = {"lib": pkg.library().exaadmm_version().decode(), "states": []}
for seed, noise in ((0, 0.0), (1, 0.02), (2, 0.2)):
    g = syn.sized_like("case13659pegase", seed=seed)
    rho = (5e1, 5e3)
    ref = oracle.OracleModel(g, *rho, nthreads=min(16, os.cpu_count() or 1))
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
    mod.info.inner = 7
    pkg.admm_update_x(env, mod)
    a_gpu, m_gpu, _ = mod.branch_counters()
    u_gpu = mod.solution.u_curr[2 * G_:].reshape(L_, 8)
    mb_gpu = mod.membuf.T
    u_ref, a_ref, m_ref, mult = oracle_branch_subset(oracle, g, ref.sol, ref.membuf, np.arange(L_), major_iter=7)
    same = (a_gpu == a_ref) & (m_gpu == m_ref)
    err = np.abs(u_gpu - u_ref).max(axis=1)
    out["states"].append({"seed": seed, "noise": noise, "n": int(L_), "count_mismatches": int((~same).sum()),
                          "bitwise_equal_u": int((u_gpu == u_ref).all(axis=1).sum()),
                          "max_err_same": float(err[same].max()), "max_err_all": float(err.max()),
                          "penalty_mismatches": int((mb_gpu[:, 26] != mult[:, 2]).sum()),
                          "max_auglag": int(a_ref.max())})
    mod.close()
env, mod = pkg.solve_acopf(os.path.join(%(root)r, "tests", "golden", "case118.m"), outer_iterlim=25, rho_pq=4e2, rho_va=4e4,
                           outer_eps=2e-5, verbose=0)
out["case118"] = {"status": mod.info.status, "outer": mod.info.outer, "cumul": mod.info.cumul, "objval": mod.info.objval}
print("RESULT " + json.dumps(out))
'''


def _run(lib_name):
    env = dict(os.environ, EXAADMM_HIP_LIB=os.path.join(ROOT, "exaadmm.jl_amd", lib_name))
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_exact_build_matches_oracle_and_bounds_the_fast_build():
    exact = _run("libexaadmm_hip_exact.so")
    fast = _run("libexaadmm_hip.so")
    assert "-DEXA_EXACT_ORACLE_ARITH" in exact["lib"] and "-DEXA_EXACT_ORACLE_ARITH" not in fast["lib"]
    report = {"exact": exact, "fast": fast}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "exact_arith_report.json"), "w"), indent=1)
    # the reference's 118-bus anchor (acopf_update_cpu.jl:174-179) holds for both arithmetics
    for r in (exact, fast):
        assert (r["case118"]["status"], r["case118"]["outer"], r["case118"]["cumul"]) == ("Solved", 20, 1232), r["case118"]
        assert abs(r["case118"]["objval"] - 129645.676) <= 1e-6 * 129645.676
    n = sum(s["n"] for s in exact["states"])
    mis_exact = sum(s["count_mismatches"] for s in exact["states"])
    mis_fast = sum(s["count_mismatches"] for s in fast["states"])
    # exact build: device libm (sin / cos / sqrt are correctly rounded or within an ulp) is the only difference
    # from the oracle -> at most 1 branch in 10 000 may take a different number of iterations, and where the
    # counts agree the solutions agree to 1e-10
    assert mis_exact <= 1e-4 * n, report
    assert all(s["max_err_same"] <= 1e-10 for s in exact["states"]), report
    # default build: measured at 0.007-0.01 % (4-6 of 61 401: profiles/archive_r01_r02/r02_branch_arith_mismatch_classes.txt,
    # r02_exact_arith_report_gpu.json; rounds 3 and 4: 5 of 61 401 = 0.8e-4); the gate is 0.015 % (9 of 61 401)
    assert mis_fast <= 1.5e-4 * n, report
    assert all(s["max_err_same"] <= 1e-7 for s in fast["states"]), report
    assert all(s["penalty_mismatches"] <= s["count_mismatches"] for s in fast["states"] + exact["states"])
