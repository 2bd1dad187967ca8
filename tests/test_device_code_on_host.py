"""Run the product's per-lane solver code (exaadmm.jl_amd/csrc/*.h), compiled for the
HOST by a test-only harness, against the oracle and the reference-generated vectors.
This is a CPU pre-check of the device code; the GPU parity tests are in test_gpu_*.py."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

HC_DIR = os.path.join(ROOT, "tests", "hostcheck")
CSRC = os.path.join(ROOT, "exaadmm.jl_amd", "csrc")
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def _p(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(HC_DIR, "libhostcheck.so")
    srcs = [os.path.join(HC_DIR, "hostcheck.cpp"), os.path.join(CSRC, "tron_device.h"),
            os.path.join(CSRC, "branch_solver.h"), os.path.join(CSRC, "genramp_solver.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma",
                               "-I", CSRC, srcs[0], "-o", so, "-lm"])
    L = C.CDLL(so)
    L.hostcheck_fgh.restype = C.c_double
    L.hostcheck_fgh.argtypes = [dp, dp, dp, C.c_double, dp, dp]
    L.hostcheck_f.restype = C.c_double
    L.hostcheck_f.argtypes = [dp, dp, dp, C.c_double]
    L.hostcheck_branch_solve.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_int, C.c_double, ip, ip, ip]
    L.hostcheck_genramp_solve.argtypes = [dp, dp, dp, dp] + [C.c_double] * 4 + [C.c_int, C.c_double, ip, ip]
    L.hostcheck_genramp_fgh.restype = C.c_double
    L.hostcheck_genramp_fgh.argtypes = [dp, dp] + [C.c_double] * 4 + [dp, dp]
    return L


def test_lane_fgh_matches_reference_outputs(hc):
    d = json.load(open(os.path.join(GOLDEN, "branch_eval_vectors.json")))
    for c in d["cases"]:
        x, p, Y = (np.array(c[k]) for k in ("x", "param", "Y"))
        g = np.zeros(6)
        A = np.zeros(36)
        f = hc.hostcheck_fgh(_p(x), _p(p), _p(Y), c["scale"], _p(g), _p(A))
        f2 = hc.hostcheck_f(_p(x), _p(p), _p(Y), c["scale"])
        assert f == f2
        H, gr = np.array(c["H"]), np.array(c["g"])
        assert abs(f - c["f"]) <= 1e-12 * max(1.0, abs(c["f"]))
        assert np.abs(g - gr).max() <= 1e-12 * max(1.0, np.abs(gr).max())
        assert np.abs(A.reshape(6, 6) - H).max() <= 1e-12 * max(1.0, np.abs(H).max())


def _random_branch(rng, cold):
    r, xx, b = rng.uniform(0.0, 0.05), rng.uniform(0.02, 0.3), rng.uniform(0, 0.4)
    tap = rng.choice([1.0, 1.0, 0.98, 1.05]) * np.exp(1j * rng.choice([0.0, 0.0, 0.05]))
    Ys = 1 / (r + 1j * xx)
    Ytt = Ys + 1j * b / 2
    Yff = Ytt / (tap * np.conj(tap))
    Yft = -Ys / np.conj(tap)
    Ytf = -Ys / tap
    Y = np.array([Yff.real, Yff.imag, Yft.real, Yft.imag, Ytt.real, Ytt.imag, Ytf.real, Ytf.imag])
    rate = rng.choice([1e3, 0.99 * rng.uniform(0.3, 4.0) ** 2])
    xl = np.array([0.9, 0.9, -2 * np.pi, -2 * np.pi, -rate, -rate])
    xu = np.array([1.1, 1.1, 2 * np.pi, 2 * np.pi, 0.0, 0.0])
    if rng.random() < 0.15:
        xl[2] = xu[2] = 0.0   # reference bus on the from side
    p = np.zeros(31)
    rho_pq, rho_va = rng.choice([(10.0, 1e3), (50.0, 5e3), (400.0, 4e4)])
    p[8:12], p[12:16] = rho_pq, rho_va
    if cold:
        w0 = 1.01
        p[16:24] = [Y[0] * w0 + Y[2] * w0, -Y[1] * w0 - Y[3] * w0, Y[4] * w0 + Y[6] * w0,
                    -Y[5] * w0 - Y[7] * w0, w0, w0, 0, 0]
        x = np.array([0.9, 0.9, 0, 0, 0, 0.0])
    else:
        vi, vj, th = rng.uniform(0.92, 1.08), rng.uniform(0.92, 1.08), rng.uniform(-0.4, 0.4)
        ti = 0.0 if xl[2] == xu[2] else rng.uniform(-0.3, 0.3)
        tj = ti - th
        cs, sn = np.cos(th), np.sin(th)
        flow = np.array([Y[0] * vi * vi + Y[2] * vi * vj * cs + Y[3] * vi * vj * sn,
                         -Y[1] * vi * vi - Y[3] * vi * vj * cs + Y[2] * vi * vj * sn,
                         Y[4] * vj * vj + Y[6] * vi * vj * cs - Y[7] * vi * vj * sn,
                         -Y[5] * vj * vj - Y[7] * vi * vj * cs - Y[6] * vi * vj * sn])
        p[16:24] = np.concatenate([flow, [vi * vi, vj * vj, ti, tj]]) + rng.normal(0, 0.02, 8)
        p[0:8] = rng.normal(0, 5.0, 8) * np.array([1, 1, 1, 1, 10, 10, 10, 10])
        p[24:26] = rng.choice([0.0, 1.0]) * rng.uniform(0, 3, 2)
        x = np.array([vi, vj, ti, tj, max(-rate, -(flow[0] ** 2 + flow[1] ** 2)),
                      max(-rate, -(flow[2] ** 2 + flow[3] ** 2))])
        x += rng.normal(0, 0.01, 6)
        x = np.minimum(xu, np.maximum(xl, x))
    p[26] = 10.0 ** rng.integers(1, 4)
    p[28] = rate
    return x, xl, xu, p, Y


@pytest.mark.parametrize("cold", [True, False])
def test_lane_branch_solve_matches_oracle(hc, oracle, cold):
    rng = np.random.default_rng(7 if cold else 11)
    OL = oracle.lib()
    n_same_counts = 0
    ncase = 300
    for _ in range(ncase):
        x, xl, xu, p, Y = _random_branch(rng, cold)
        scale = float(rng.choice([1e-4, 1e-5]))
        xo, po = x.copy(), p.copy()
        ito, mio = C.c_int(), C.c_int()
        OL.oracle_branch_solve(_p(xo), _p(xl), _p(xu), _p(po), _p(Y), scale, 50, 1e8,
                               C.cast(C.byref(ito), oracle.c_int_p), C.cast(C.byref(mio), oracle.c_int_p))
        xh, ph = x.copy(), p.copy()
        ith, mih, tk = C.c_int(), C.c_int(), C.c_int()
        hc.hostcheck_branch_solve(_p(xh), _p(xl), _p(xu), _p(ph), _p(Y), scale, 50, 1e8,
                                  C.byref(ith), C.byref(mih), C.byref(tk))
        # the lane code evaluates f/g/H through a different (cheaper) factorisation of the same
        # mathematics, so iterates agree to rounding-amplified-by-conditioning, not bitwise
        assert np.abs(xh - xo).max() <= 1e-7, (xh, xo)
        assert np.abs(ph[24:27] - po[24:27]).max() <= 1e-6 * max(1.0, np.abs(po[24:27]).max())
        n_same_counts += (ith.value == ito.value and mih.value == mio.value)
        assert tk.value >= mih.value - ith.value
    assert n_same_counts >= 0.97 * ncase


def test_lane_generator_ramp_solve_matches_oracle(hc, oracle):
    """3-variable generator ramp subproblem of the multi-period model: lane code
    (TRON with N = 3) against the oracle's restatement of mpacopf_auglag_generator_kernel_cpu.jl."""
    rng = np.random.default_rng(23)
    oracle._mp_lib()
    OL = oracle.lib()
    same = 0
    ncase = 400
    for _ in range(ncase):
        pmax = rng.uniform(0.5, 5.0)
        pmin = rng.choice([0.0, 0.1 * pmax])
        ramp = 0.02 * pmax
        xl = np.array([pmin, pmin, -ramp])
        xu = np.array([pmax, pmax, ramp])
        x = np.array([rng.uniform(pmin, pmax), rng.uniform(pmin, pmax), rng.uniform(-ramp, ramp)])
        rho = float(rng.choice([10.0, 50.0, 400.0]))
        param = np.array([rng.normal(0, 5), rng.normal(0, 5), rho, rho, rng.uniform(pmin, pmax),
                          rng.uniform(pmin, pmax), rng.choice([0.0, rng.normal(0, 20)]), 10.0 ** rng.integers(1, 5)])
        c2, c1, c0, M = rng.uniform(0.0, 0.2), rng.uniform(1, 40), rng.uniform(0, 500), 100.0
        xo, po = x.copy(), param.copy()
        ito, mio = C.c_int(), C.c_int()
        OL.oracle_genramp_solve(_p(xo), _p(xl), _p(xu), _p(po), 1.0, c2, c1, c0, M, 50, 1e8,
                                C.cast(C.byref(ito), oracle.c_int_p), C.cast(C.byref(mio), oracle.c_int_p))
        xh, ph = x.copy(), param.copy()
        ith, mih = C.c_int(), C.c_int()
        hc.hostcheck_genramp_solve(_p(xh), _p(xl), _p(xu), _p(ph), c2, c1, c0, M, 50, 1e8, C.byref(ith), C.byref(mih))
        assert np.abs(xh - xo).max() <= 1e-9 * max(1.0, pmax), (xh, xo)
        assert np.abs(ph[6:8] - po[6:8]).max() <= 1e-7 * max(1.0, np.abs(po[6:8]).max())
        same += (ith.value == ito.value and mih.value == mio.value)
    assert same >= 0.97 * ncase


def test_small_angle_sincos_of_the_device_build_is_libm_to_one_ulp(hc):
    """The default device build evaluates sin / cos of the angle difference with the two kernel polynomials for |t| <= pi/4
    (branch_solver.h sincos_quarter_pi; library call beyond).  Against libm on 2 million points of [-pi/4, pi/4] incl. the end
    points, tiny arguments and 0: at most one unit in the last place apart, exact at 0, sin odd and cos even to the bit."""
    rng = np.random.default_rng(7)
    q = np.pi / 4
    t = np.concatenate([rng.uniform(-q, q, 2_000_000), np.linspace(-q, q, 100_001), 10.0 ** rng.uniform(-300, -1, 20_000) * rng.choice([-1.0, 1.0], 20_000),
                        [0.0, -0.0, q, -q, np.nextafter(q, 0), 5e-324, 1e-160]])
    t = np.ascontiguousarray(t[np.abs(t) <= 0.78539816339744830962])
    sn, cs = np.empty_like(t), np.empty_like(t)
    hc.hostcheck_sincos_quarter_pi.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    hc.hostcheck_sincos_quarter_pi(_p(t), t.size, _p(sn), _p(cs))
    rs, rc = np.sin(t), np.cos(t)
    ulp_s = np.abs(sn - rs) / np.spacing(np.abs(rs) + 5e-324)
    ulp_c = np.abs(cs - rc) / np.spacing(np.abs(rc))
    assert ulp_s.max() <= 1.0 and ulp_c.max() <= 1.0, (ulp_s.max(), ulp_c.max())
    assert (ulp_s > 0).mean() < 0.2 and (ulp_c > 0).mean() < 0.2      # and mostly identical
    z = t == 0.0
    assert np.all(sn[z] == 0.0) and np.all(cs[z] == 1.0)
    sn2, cs2 = np.empty_like(t), np.empty_like(t)
    tm = np.ascontiguousarray(-t)
    hc.hostcheck_sincos_quarter_pi(_p(tm), t.size, _p(sn2), _p(cs2))
    assert np.array_equal(sn2, -sn) and np.array_equal(cs2, cs)


def test_lane_code_gives_up_quickly_on_non_finite_inputs(hc):
    """NaN in the consensus targets must not make a lane spin through its guarded loops
    (50 x 500 evaluations with 1000-trip line searches = seconds on a GPU): one tick, then out,
    with a finite point returned."""
    rng = np.random.default_rng(3)
    x, xl, xu, p, Y = _random_branch(rng, cold=False)
    p[18] = np.nan
    ith, mih, tk = C.c_int(), C.c_int(), C.c_int()
    hc.hostcheck_branch_solve(_p(x), _p(xl), _p(xu), _p(p), _p(Y), 1e-4, 50, 1e8, C.byref(ith), C.byref(mih), C.byref(tk))
    assert tk.value <= 2
    assert np.all(np.isfinite(x)) and np.all(x >= xl) and np.all(x <= xu)
    xg = np.array([0.5, 0.4, 0.01]); xlg = np.array([0.0, 0.0, -0.02]); xug = np.array([1.0, 1.0, 0.02])
    pg = np.array([0.0, 0.0, 400.0, 400.0, np.nan, 0.4, 0.0, 10.0])
    hc.hostcheck_genramp_solve(_p(xg), _p(xlg), _p(xug), _p(pg), 0.1, 10.0, 0.0, 100.0, 50, 1e8, C.byref(ith), C.byref(mih))
    assert np.all(np.isfinite(xg)) and mih.value <= 2


def test_lane_generator_ramp_callbacks_match_reference_outputs(hc):
    """f, grad f, Hessian of the generator ramp subproblem against the reference's own callbacks
    executed at fixture-generation time (tests/golden/gen_genramp_eval_vectors.py); the lane code
    fixes scale = 1 as the reference's call site does (mpacopf_admm_update_x_cpu.jl:16)."""
    d = json.load(open(os.path.join(GOLDEN, "genramp_eval_vectors.json")))
    n = 0
    for c in d["cases"]:
        if c["scale"] != 1.0:
            continue
        n += 1
        x, p = np.array(c["x"]), np.array(c["param"])
        g, A = np.zeros(3), np.zeros(9)
        f = hc.hostcheck_genramp_fgh(_p(x), _p(p), c["c2"], c["c1"], c["c0"], c["baseMVA"], _p(g), _p(A))
        assert abs(f - c["f"]) <= 1e-13 * max(1.0, abs(c["f"]))
        assert np.abs(g - c["g"]).max() <= 1e-13 * max(1.0, np.abs(c["g"]).max())
        assert np.abs(A.reshape(3, 3) - np.array(c["H"])).max() <= 1e-13 * max(1.0, np.abs(c["H"]).max())
    assert n >= 8


@pytest.fixture(scope="module")
def hc_exact():
    """the same lane code compiled with -DEXA_EXACT_ORACLE_ARITH (the CPU path's operation order)"""
    so = os.path.join(HC_DIR, "libhostcheck_exact.so")
    srcs = [os.path.join(HC_DIR, "hostcheck.cpp"), os.path.join(CSRC, "tron_device.h"),
            os.path.join(CSRC, "branch_solver.h"), os.path.join(CSRC, "genramp_solver.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s_) > os.path.getmtime(so) for s_ in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma",
                               "-DEXA_EXACT_ORACLE_ARITH", "-I", CSRC, srcs[0], "-o", so, "-lm"])
    L = C.CDLL(so)
    L.hostcheck_branch_solve.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_int, C.c_double, ip, ip, ip]
    L.hostcheck_genramp_solve.argtypes = [dp, dp, dp, dp] + [C.c_double] * 4 + [C.c_int, C.c_double, ip, ip]
    return L


@pytest.mark.parametrize("cold", [True, False])
def test_exact_arith_build_is_bitwise_the_oracle_on_branches(hc_exact, oracle, cold):
    """-DEXA_EXACT_ORACLE_ARITH: same libm on the host, so the lane code must reproduce the oracle BIT FOR BIT
    (x, multipliers, penalty, iteration counts) on every problem -- this pins the flattened tick loop, the
    masked (uncompacted) face loop and every TRON routine of tron_device.h to the CPU restatement; what is left
    to the default build is arithmetic only (FMA, reciprocal pivots, Cholesky shortcut, re-derived f/g/H), whose
    effect tests/tools/classify_mismatches.py counts (profiles/archive_r01_r02/r02_branch_arith_mismatch_classes.txt)."""
    rng = np.random.default_rng(70 if cold else 110)
    OL = oracle.lib()
    for _ in range(400):
        x, xl, xu, p, Y = _random_branch(rng, cold)
        scale = float(rng.choice([1e-4, 1e-5]))
        xo, po = x.copy(), p.copy()
        ito, mio = C.c_int(), C.c_int()
        OL.oracle_branch_solve(_p(xo), _p(xl), _p(xu), _p(po), _p(Y), scale, 50, 1e8,
                               C.cast(C.byref(ito), oracle.c_int_p), C.cast(C.byref(mio), oracle.c_int_p))
        xh, ph = x.copy(), p.copy()
        ith, mih, tk = C.c_int(), C.c_int(), C.c_int()
        hc_exact.hostcheck_branch_solve(_p(xh), _p(xl), _p(xu), _p(ph), _p(Y), scale, 50, 1e8,
                                        C.byref(ith), C.byref(mih), C.byref(tk))
        assert (ith.value, mih.value) == (ito.value, mio.value)
        assert np.array_equal(xh, xo) and np.array_equal(ph[24:27], po[24:27])


def test_exact_arith_build_is_bitwise_the_oracle_on_ramp_subproblems(hc_exact, oracle):
    rng = np.random.default_rng(230)
    oracle._mp_lib()
    OL = oracle.lib()
    for _ in range(600):
        pmax = rng.uniform(0.5, 5.0)
        pmin = rng.choice([0.0, 0.1 * pmax])
        ramp = 0.02 * pmax
        xl = np.array([pmin, pmin, -ramp])
        xu = np.array([pmax, pmax, ramp])
        x = np.array([rng.uniform(pmin, pmax), rng.uniform(pmin, pmax), rng.uniform(-ramp, ramp)])
        rho = float(rng.choice([10.0, 50.0, 400.0]))
        param = np.array([rng.normal(0, 5), rng.normal(0, 5), rho, rho, rng.uniform(pmin, pmax),
                          rng.uniform(pmin, pmax), rng.choice([0.0, rng.normal(0, 20)]), 10.0 ** rng.integers(1, 5)])
        c2, c1, c0, M = rng.uniform(0.0, 0.2), rng.uniform(1, 40), rng.uniform(0, 500), 100.0
        xo, po = x.copy(), param.copy()
        ito, mio = C.c_int(), C.c_int()
        OL.oracle_genramp_solve(_p(xo), _p(xl), _p(xu), _p(po), 1.0, c2, c1, c0, M, 50, 1e8,
                                C.cast(C.byref(ito), oracle.c_int_p), C.cast(C.byref(mio), oracle.c_int_p))
        xh, ph = x.copy(), param.copy()
        ith, mih = C.c_int(), C.c_int()
        hc_exact.hostcheck_genramp_solve(_p(xh), _p(xl), _p(xu), _p(ph), c2, c1, c0, M, 50, 1e8, C.byref(ith), C.byref(mih))
        assert (ith.value, mih.value) == (ito.value, mio.value)
        assert np.array_equal(xh, xo) and np.array_equal(ph, po)


def test_penalty_power_lookup_by_exponent(hc):
    """mu_pow (branch_solver.h): the table entry of mu = 10^k, k = 1..8, is found from mu's binary exponent and confirmed by
    comparison; every other mu -- neighbours of the powers of ten in the same binade, other binades, 10^0, 10^9, tiny, huge,
    zero, infinities -- must take pow()."""
    hc.hostcheck_mu_pow.restype = C.c_double
    hc.hostcheck_mu_pow.argtypes = [C.c_double, C.c_int, ip]
    hit = C.c_int()
    for k in range(1, 9):
        mu = float(10 ** k)
        assert hc.hostcheck_mu_pow(mu, 0, C.byref(hit)) == -float(k) and hit.value == 1
        assert hc.hostcheck_mu_pow(mu, 1, C.byref(hit)) == -float(k) - 100.0 and hit.value == 1
        for other in (np.nextafter(mu, 0.0), np.nextafter(mu, np.inf), mu * 1.5, mu * 0.75, 2.0 ** np.floor(np.log2(mu))):
            for nine in (0, 1):
                r = hc.hostcheck_mu_pow(float(other), nine, C.byref(hit))
                assert hit.value == 0 and r == float(other) ** (0.9 if nine else 0.1)
    for other in (1.0, 1e9, 1e-3, 5e-324, 1e300, 0.0, np.inf, 3.0, 2.0 ** 27, 2.0 ** 2):
        for nine in (0, 1):
            r = hc.hostcheck_mu_pow(float(other), nine, C.byref(hit))
            assert hit.value == 0 and r == float(other) ** (0.9 if nine else 0.1)
