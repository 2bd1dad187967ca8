"""TRON primitives of the CPU restatement (oracle/tron_dense.c) against independent computations.

ExaTron.jl -- the package the reference calls for `dtron` and friends (acopf_tron_linelimit_kernel.jl:103-122,
acopf_auglag_linelimit_kernel_cpu.jl:104-116) -- is not under /root/reference, so there are no upstream unit
vectors to pin these routines to.  What can be checked without them is that each routine computes what the
published algorithm (Lin & More' 1999; MINPACK-2 dicfs/dcauchy/dtrpcg/dprsrch/dspcg/dtron) says it computes:
numpy/scipy linear algebra for the factorisation and the CG, the defining inequalities for the Cauchy point and
the projected search, and brute-force KKT enumeration / scipy.optimize for whole solves.  The device code is
pinned to this restatement bit for bit (tests/test_device_code_on_host.py, exact-arithmetic build), so these
checks reach the HIP path too.
"""
import ctypes as C
import itertools

import numpy as np
import pytest
import scipy.linalg
import scipy.optimize

from oracle import oracle as O

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def P(a):
    return a.ctypes.data_as(dp)


class Options(C.Structure):
    _fields_ = [("max_feval", C.c_int), ("max_minor", C.c_int), ("gtol", C.c_double), ("frtol", C.c_double),
                ("fatol", C.c_double), ("fmin", C.c_double), ("cgtol", C.c_double)]


EVAL_F = C.CFUNCTYPE(C.c_double, dp, C.c_void_p)
EVAL_GH = C.CFUNCTYPE(None, dp, dp, dp, C.c_void_p)


@pytest.fixture(scope="module")
def T():
    L = O.lib()
    L.tron_dnrm2.argtypes = [C.c_int, dp]
    L.tron_dnrm2.restype = C.c_double
    L.tron_dgpnorm.argtypes = [C.c_int, dp, dp, dp, dp]
    L.tron_dgpnorm.restype = C.c_double
    L.tron_dbreakpt.argtypes = [C.c_int, dp, dp, dp, dp, ip, dp, dp]
    L.tron_dgpstep.argtypes = [C.c_int, dp, dp, dp, C.c_double, dp, dp]
    L.tron_dtrqsol.argtypes = [C.c_int, dp, dp, C.c_double]
    L.tron_dtrqsol.restype = C.c_double
    L.tron_dicfs.argtypes = [C.c_int, C.c_double, dp, dp]
    L.tron_dicfs.restype = C.c_int
    L.tron_dcauchy.argtypes = [C.c_int, dp, dp, dp, dp, dp, C.c_double, dp, dp]
    L.tron_dtrpcg.argtypes = [C.c_int, dp, dp, C.c_double, dp, C.c_double, C.c_double, C.c_int, dp, ip]
    L.tron_dtrpcg.restype = C.c_int
    L.tron_dprsrch.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp]
    L.tron_dspcg.argtypes = [C.c_int, dp, dp, dp, dp, dp, C.c_double, C.c_double, dp, C.c_int, ip]
    L.tron_dspcg.restype = C.c_int
    L.tron_default_options.argtypes = [C.POINTER(Options)]
    L.tron_solve.argtypes = [C.c_int, dp, dp, dp, EVAL_F, EVAL_GH, C.c_void_p, C.POINTER(Options), ip, ip]
    L.tron_solve.restype = C.c_int
    return L


def spd(rng, n, cond=1e3):
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.exp(rng.uniform(0.0, np.log(cond), n))
    a = (q * ev) @ q.T
    return np.asfortranarray(0.5 * (a + a.T))


def sym(rng, n):
    a = rng.standard_normal((n, n))
    return np.asfortranarray(0.5 * (a + a.T))


def box_problem(rng, n, active=0.4):
    """x strictly inside or on the faces of a random box."""
    xl = -rng.uniform(0.2, 2.0, n)
    xu = rng.uniform(0.2, 2.0, n)
    x = rng.uniform(xl, xu)
    for i in range(n):
        r = rng.uniform()
        if r < active / 2:
            x[i] = xl[i]
        elif r < active:
            x[i] = xu[i]
    return x, xl, xu


def q_of(A, g, s):
    return 0.5 * s @ A @ s + g @ s


# ---- norms, break points, projected step -------------------------------------------------------------------------
def test_norms(T):
    rng = np.random.default_rng(0)
    for n in range(1, 7):
        for _ in range(20):
            x, xl, xu = box_problem(rng, n)
            g = rng.standard_normal(n)
            assert T.tron_dnrm2(n, P(g)) == pytest.approx(np.linalg.norm(g), rel=1e-15)
            # projected gradient, infinity norm: components that push against an active bound do not count;
            # a variable whose bounds coincide never counts
            pg = np.where(x == xl, np.minimum(g, 0.0), np.where(x == xu, np.maximum(g, 0.0), g))
            assert T.tron_dgpnorm(n, P(x), P(xl), P(xu), P(g)) == np.abs(pg).max()
    x = np.array([1.0, 0.0])
    lo = np.array([1.0, -1.0])
    hi = np.array([1.0, 1.0])
    g = np.array([5.0, 0.25])
    assert T.tron_dgpnorm(2, P(x), P(lo), P(hi), P(g)) == 0.25


def test_breakpoints_and_projected_step(T):
    rng = np.random.default_rng(1)
    for n in range(1, 7):
        for _ in range(40):
            x, xl, xu = box_problem(rng, n)
            w = rng.standard_normal(n)
            w[rng.uniform(size=n) < 0.15] = 0.0
            nb, bmin, bmax = C.c_int(), C.c_double(), C.c_double()
            T.tron_dbreakpt(n, P(x), P(xl), P(xu), P(w), C.byref(nb), C.byref(bmin), C.byref(bmax))
            # break point of coordinate i: the step length at which x + t w leaves the box through that coordinate
            br = []
            for i in range(n):
                if w[i] < 0 and x[i] > xl[i]:
                    br.append((xl[i] - x[i]) / w[i])
                elif w[i] > 0 and x[i] < xu[i]:
                    br.append((xu[i] - x[i]) / w[i])
            assert nb.value == len(br)
            assert bmin.value == (min(br) if br else 0.0)
            assert bmax.value == (max(br) if br else 0.0)
            for alpha in (0.0, 0.3, 1.0, 7.0):
                s = np.empty(n)
                T.tron_dgpstep(n, P(x), P(xl), P(xu), alpha, P(w), P(s))
                ref = np.clip(x + alpha * w, xl, xu) - x
                np.testing.assert_allclose(s, ref, rtol=0, atol=4e-16)
                inside = (x + alpha * w >= xl) & (x + alpha * w <= xu)
                assert np.array_equal(s[inside], (alpha * w)[inside])


def test_trust_region_boundary_root(T):
    rng = np.random.default_rng(2)
    for n in range(1, 7):
        for _ in range(50):
            x = rng.standard_normal(n)
            p = rng.standard_normal(n)
            delta = np.linalg.norm(x) * rng.uniform(1.0, 3.0) + 1e-3
            sigma = T.tron_dtrqsol(n, P(x), P(p), delta)
            assert sigma >= 0.0
            assert np.linalg.norm(x + sigma * p) == pytest.approx(delta, rel=1e-12)
    z = np.zeros(3)
    assert T.tron_dtrqsol(3, P(z), P(z), 1.0) == 0.0


# ---- dicfs: scaled, shifted Cholesky -----------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
def test_dicfs_is_the_cholesky_factor_of_a_positive_definite_matrix(T, n):
    rng = np.random.default_rng(10 + n)
    for _ in range(30):
        A = spd(rng, n, cond=10 ** rng.uniform(0, 8))
        L = np.zeros((n, n), order="F")
        nb = T.tron_dicfs(n, 0.0, P(A), P(L))
        assert nb == 1                                    # no shift tried
        assert np.array_equal(np.triu(L, 1), np.zeros((n, n)))
        ref = np.linalg.cholesky(A)
        np.testing.assert_allclose(L, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
        np.testing.assert_allclose(L @ L.T, A, rtol=0, atol=1e-13 * np.abs(A).max())


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6])
def test_dicfs_shifts_an_indefinite_matrix_along_its_scaled_diagonal(T, n):
    """A + alpha*D = L L', D = diag(||A(:,i)||_2) (the scaling of dicfs), alpha > 0 the smallest shift of the
    doubling sequence that makes the scaled matrix positive definite."""
    rng = np.random.default_rng(20 + n)
    seen = 0
    for _ in range(60):
        A = sym(rng, n)
        if np.linalg.eigvalsh(A).min() > 0:
            continue
        seen += 1
        L = np.zeros((n, n), order="F")
        T.tron_dicfs(n, 0.0, P(A), P(L))
        E = L @ L.T - A
        off = E - np.diag(np.diag(E))
        assert np.abs(off).max() <= 1e-12 * max(1.0, np.abs(A).max())
        colnorm = np.linalg.norm(A, axis=0)
        alpha = np.diag(E) / colnorm
        assert alpha.min() > 0
        np.testing.assert_allclose(alpha, alpha[0], rtol=1e-9)
        a = alpha[0]
        # the scaled matrix S A S + a I is positive definite, and the shift is in the routine's sequence
        # max(initial, 1e-3) * 2^k
        S = np.diag(1.0 / np.sqrt(colnorm))
        assert np.linalg.eigvalsh(S @ A @ S + a * np.eye(n)).min() > 0
        a0 = max(1e-3, max(0.0, (-np.diag(A) / colnorm).max()))
        k = np.log2(a / a0)
        assert abs(k - round(k)) < 1e-9 and round(k) >= 0
        # ... and the previous member of the sequence was not enough (or the first was taken)
        if round(k) > 0:
            assert np.linalg.eigvalsh(S @ A @ S + 0.5 * a * np.eye(n)).min() <= 1e-12
    assert seen >= 20


# ---- dtrpcg ------------------------------------------------------------------------------------------------------
def trpcg(T, A, g, delta, L, tol, stol, itermax):
    n = len(g)
    w = np.zeros(n)
    it = C.c_int()
    info = T.tron_dtrpcg(n, P(A), P(g), delta, P(L), tol, stol, itermax, P(w), C.byref(it))
    return info, it.value, w


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
def test_trpcg_with_the_exact_factor_is_one_newton_step(T, n):
    rng = np.random.default_rng(30 + n)
    for _ in range(20):
        A = spd(rng, n)
        L = np.asfortranarray(np.linalg.cholesky(A))
        g = rng.standard_normal(n)
        info, it, w = trpcg(T, A, g, 1e6, L, 1e-10 * np.linalg.norm(g), 0.0, n)
        assert (info, it) == (1, 1)
        s = scipy.linalg.solve_triangular(L.T, w, lower=False)     # back to the original variables
        np.testing.assert_allclose(s, -np.linalg.solve(A, g), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6])
def test_trpcg_is_conjugate_gradients_truncated_at_the_boundary(T, n):
    """Identity preconditioner: the iterates are those of plain CG on A w = -g (finite termination in n steps);
    with a small radius the result lies on the boundary, at least as good as the steepest-descent boundary point."""
    rng = np.random.default_rng(40 + n)
    I = np.asfortranarray(np.eye(n))
    for _ in range(20):
        A = spd(rng, n, cond=50.0)
        g = rng.standard_normal(n)
        info, it, w = trpcg(T, A, g, 1e6, I, 1e-9 * np.linalg.norm(g), 0.0, n)
        assert info in (1, 5) and it <= n
        assert np.linalg.norm(A @ w + g) <= 1e-7 * np.linalg.norm(g)
        # k steps of CG minimise q over the Krylov space of dimension k: compare with a direct solve on that space
        for k in range(1, n):
            info, it, wk = trpcg(T, A, g, 1e6, I, 0.0, 0.0, k)
            K = np.stack([np.linalg.matrix_power(A, j) @ g for j in range(k)], axis=1)
            Q, _ = np.linalg.qr(K)
            ref = -Q @ np.linalg.solve(Q.T @ A @ Q, Q.T @ g)
            assert it == k
            np.testing.assert_allclose(wk, ref, rtol=1e-6, atol=1e-9 * np.linalg.norm(ref))
        newton = np.linalg.norm(np.linalg.solve(A, g))
        delta = 0.3 * newton
        info, it, w = trpcg(T, A, g, delta, I, 1e-12, 0.0, n)
        assert info == 4
        assert np.linalg.norm(w) == pytest.approx(delta, rel=1e-12)
        sd = -delta * g / np.linalg.norm(g)
        assert q_of(A, g, w) <= q_of(A, g, sd) + 1e-14


def test_trpcg_follows_negative_curvature_to_the_boundary(T):
    rng = np.random.default_rng(50)
    for n in range(2, 7):
        I = np.asfortranarray(np.eye(n))
        for _ in range(20):
            A = sym(rng, n) - 2.0 * np.eye(n)
            A = np.asfortranarray(A)
            if np.linalg.eigvalsh(A).max() > 0:      # want a concave model along -g at least sometimes
                pass
            g = rng.standard_normal(n)
            delta = rng.uniform(0.1, 2.0)
            info, it, w = trpcg(T, A, g, delta, I, 1e-12, 0.0, n)
            assert info in (3, 4)
            assert np.linalg.norm(w) == pytest.approx(delta, rel=1e-12)
            assert q_of(A, g, w) < 0.0
        z = np.zeros(n)
        info, it, w = trpcg(T, A, z, 1.0, I, 1e-12, 0.0, n)
        assert (info, it) == (1, 0) and not w.any()


# ---- Cauchy point and projected search: the defining inequalities --------------------------------------------------
@pytest.mark.parametrize("definite", [True, False])
def test_cauchy_step_satisfies_sufficient_decrease_inside_the_region(T, definite):
    rng = np.random.default_rng(60 + definite)
    mu0 = 0.01
    for n in range(1, 7):
        for _ in range(60):
            x, xl, xu = box_problem(rng, n)
            A = spd(rng, n) if definite else sym(rng, n)
            g = rng.standard_normal(n) * 10 ** rng.uniform(-2, 2)
            pg = np.clip(x - g, xl, xu) - x
            if not pg.any():
                continue
            delta = np.linalg.norm(g) * 10 ** rng.uniform(-2, 1)
            for alpha0 in (1.0, 1e-3, 50.0):
                alpha = C.c_double(alpha0)
                s = np.empty(n)
                T.tron_dcauchy(n, P(x), P(xl), P(xu), P(A), P(g), delta, C.byref(alpha), P(s))
                a = alpha.value
                assert a > 0.0
                np.testing.assert_allclose(s, np.clip(x - a * g, xl, xu) - x, rtol=0, atol=4e-16)
                assert np.linalg.norm(s) <= delta * (1 + 1e-15)
                assert q_of(A, g, s) <= mu0 * (g @ s) + 1e-15 * abs(g @ s)
                assert g @ s < 0.0
                # the step is not needlessly short: either the trial ten times longer fails one of the two tests,
                # or it is the same step (everything already at a bound), or it was never tried because the
                # search started from a failing alpha0 (interpolation stops at the first success)
                s10 = np.clip(x - 10 * a * g, xl, xu) - x
                started_ok = (np.linalg.norm(np.clip(x - alpha0 * g, xl, xu) - x) <= delta and
                              q_of(A, g, np.clip(x - alpha0 * g, xl, xu) - x) < mu0 * (g @ (np.clip(x - alpha0 * g, xl, xu) - x)))
                if started_ok and not np.allclose(s10, s, rtol=0, atol=1e-15):
                    fails = (np.linalg.norm(s10) > delta) or (q_of(A, g, s10) >= mu0 * (g @ s10))
                    assert fails


def test_projected_search_returns_a_feasible_point_with_sufficient_decrease(T):
    rng = np.random.default_rng(70)
    mu0 = 0.01
    for n in range(1, 7):
        for _ in range(80):
            x, xl, xu = box_problem(rng, n, active=0.0)
            A = spd(rng, n)
            g = rng.standard_normal(n)
            # a (possibly shortened) Newton direction, as dtrpcg hands over: q decreases along it up to length one,
            # which is what the routine relies on when the whole step stays inside the box (it is then not tested)
            w = -np.linalg.solve(A, g) * rng.uniform(0.3, 1.0)
            x0 = x.copy()
            wio = w.copy()
            T.tron_dprsrch(n, P(x), P(xl), P(xu), P(A), P(g), P(wio))
            assert np.all(x >= xl) and np.all(x <= xu)
            np.testing.assert_allclose(x - x0, wio, rtol=0, atol=1e-14)
            assert q_of(A, g, wio) <= mu0 * (g @ wio) + 1e-15
            # the full step is taken when it is acceptable
            full = np.clip(x0 + w, xl, xu) - x0
            if q_of(A, g, full) <= mu0 * (g @ full):
                np.testing.assert_allclose(wio, full, rtol=0, atol=1e-15)


def test_subspace_cg_improves_on_the_cauchy_step_and_stays_feasible(T):
    rng = np.random.default_rng(80)
    for n in range(1, 7):
        for _ in range(60):
            x, xl, xu = box_problem(rng, n)
            A = spd(rng, n) if rng.uniform() < 0.5 else sym(rng, n)
            g = rng.standard_normal(n)
            if not (np.clip(x - g, xl, xu) - x).any():
                continue
            delta = np.linalg.norm(g) * 10 ** rng.uniform(-1, 1)
            alpha = C.c_double(1.0)
            s = np.empty(n)
            T.tron_dcauchy(n, P(x), P(xl), P(xu), P(A), P(g), delta, C.byref(alpha), P(s))
            sc = s.copy()
            x0 = x.copy()
            it = C.c_int()
            info = T.tron_dspcg(n, P(x), P(xl), P(xu), P(A), P(g), delta, 0.1, P(s), n, C.byref(it))
            assert info in (1, 2, 3)
            assert np.all(x >= xl) and np.all(x <= xu)
            np.testing.assert_allclose(x - x0, s, rtol=0, atol=1e-13)
            assert q_of(A, g, s) <= q_of(A, g, sc) + 1e-13 * max(1.0, abs(q_of(A, g, sc)))
            # variables the Cauchy step put on a bound stay there (the face can only shrink)
            xc = np.clip(x0 + sc, xl, xu)
            stuck = (xc == xl) | (xc == xu)
            assert np.array_equal(x[stuck], xc[stuck])
            if info == 1:
                # converged on the final face: reduced gradient small relative to the free part of g
                free = (x > xl) & (x < xu)
                if free.any():
                    r = (A @ s + g)[free]
                    assert np.linalg.norm(r) <= 0.1 * np.linalg.norm(g[free]) * (1 + 1e-12)


# ---- whole solves --------------------------------------------------------------------------------------------------
def solve(T, n, x0, xl, xu, f, gh, **kw):
    opt = Options()
    T.tron_default_options(C.byref(opt))
    for k, v in kw.items():
        setattr(opt, k, v)
    x = np.array(x0, dtype=float)

    def cf(xp, _):
        return f(np.ctypeslib.as_array(xp, (n,)).copy())

    def cgh(xp, gp, Ap, _):
        g, A = gh(np.ctypeslib.as_array(xp, (n,)).copy())
        np.ctypeslib.as_array(gp, (n,))[:] = g
        np.ctypeslib.as_array(Ap, (n * n,))[:] = np.asarray(A).reshape(-1, order="F")

    minor, nfev = C.c_int(), C.c_int()
    status = T.tron_solve(n, P(x), P(xl), P(xu), EVAL_F(cf), EVAL_GH(cgh), None, C.byref(opt),
                          C.byref(minor), C.byref(nfev))
    return status, x, minor.value, nfev.value


def box_qp_by_enumeration(A, g, xl, xu):
    """Exact minimiser of a strictly convex QP over a box: try every assignment of {lower, free, upper}."""
    n = len(g)
    best = None
    for pat in itertools.product((0, 1, 2), repeat=n):
        pat = np.array(pat)
        x = np.where(pat == 0, xl, np.where(pat == 2, xu, 0.0))
        fr = pat == 1
        if fr.any():
            rhs = -(g[fr] + A[np.ix_(fr, ~fr)] @ x[~fr])
            x[fr] = np.linalg.solve(A[np.ix_(fr, fr)], rhs)
            if np.any(x[fr] < xl[fr]) or np.any(x[fr] > xu[fr]):
                continue
        grad = A @ x + g
        if np.any(grad[pat == 0] < -1e-12) or np.any(grad[pat == 2] > 1e-12):
            continue
        v = q_of(A, g, x)
        if best is None or v < best[0]:
            best = (v, x)
    return best[1]


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
def test_tron_solves_convex_box_qps_to_the_enumerated_minimiser(T, n):
    rng = np.random.default_rng(90 + n)
    for _ in range(12):
        A = spd(rng, n, cond=100.0)
        g = rng.standard_normal(n) * 3.0
        x0, xl, xu = box_problem(rng, n)
        status, x, minor, nfev = solve(T, n, x0, xl, xu, lambda z: q_of(A, g, z), lambda z: (A @ z + g, A))
        assert status == 0
        ref = box_qp_by_enumeration(A, g, xl, xu)
        np.testing.assert_allclose(x, ref, rtol=0, atol=2e-6)
        assert T.tron_dgpnorm(n, P(x), P(xl), P(xu), P(A @ x + g)) <= 1e-6
        assert minor <= 3 * n + 3


def rosen(z):
    return float(np.sum(100.0 * (z[1:] - z[:-1] ** 2) ** 2 + (1 - z[:-1]) ** 2))


def rosen_gh(z):
    return scipy.optimize.rosen_der(z), scipy.optimize.rosen_hess(z)


def test_tron_on_the_bounded_rosenbrock_function(T):
    """Non-convex, indefinite Hessians on the way: exercises the shift in dicfs, negative curvature in dtrpcg,
    the radius update and rejected steps.  Unbounded: the known minimiser (1,...,1); bounded: scipy's answer."""
    for n in (2, 4, 6):
        lo = np.full(n, -10.0)
        hi = np.full(n, 10.0)
        status, x, minor, nfev = solve(T, n, np.full(n, -1.2), lo, hi, rosen, rosen_gh)
        assert status == 0
        np.testing.assert_allclose(x, np.ones(n), rtol=0, atol=1e-5)
        assert nfev >= minor                      # (some steps were rejected or it would be equal)
        hi = np.full(n, 0.8)                       # the minimiser is now on the boundary
        hi[0] = 0.5
        status, x, minor, nfev = solve(T, n, np.full(n, -1.2), lo, hi, rosen, rosen_gh)
        assert status == 0
        ref = scipy.optimize.minimize(rosen, np.full(n, -1.2), jac=scipy.optimize.rosen_der, method="L-BFGS-B",
                                      bounds=list(zip(lo, hi)), options={"ftol": 1e-15, "gtol": 1e-10, "maxiter": 5000})
        assert rosen(x) <= ref.fun + 1e-9
        np.testing.assert_allclose(x, ref.x, rtol=0, atol=2e-4)
        g = scipy.optimize.rosen_der(x)
        assert T.tron_dgpnorm(n, P(x), P(lo), P(hi), P(g)) <= 1e-6


def test_tron_respects_its_iteration_limits(T):
    n = 6
    lo = np.full(n, -10.0)
    hi = np.full(n, 10.0)
    status, x, minor, nfev = solve(T, n, np.full(n, -1.2), lo, hi, rosen, rosen_gh, max_minor=3)
    assert status == 1 and minor == 3
    status, x, minor, nfev = solve(T, n, np.full(n, -1.2), lo, hi, rosen, rosen_gh, max_feval=4)
    assert nfev == 4
    # a stationary start is recognised after the first model step (the driver tests the projected gradient at NEWX
    # and the relative-reduction test inside dtron)
    A = np.eye(3)
    z = np.zeros(3)
    status, x, minor, nfev = solve(T, 3, z, -np.ones(3), np.ones(3), lambda v: 0.5 * v @ v, lambda v: (v, A))
    assert status == 0 and not x.any() and minor <= 2
