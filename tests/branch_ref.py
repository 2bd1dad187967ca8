"""Test helper: the oracle's single-branch solve (oracle_branch_solve) applied to a SUBSET of the
branches of an ADMM state given in the reference layout.  Keeps the CPU side of the at-scale GPU
parity tests to seconds (a full oracle x-update over 1M branches takes a minute)."""
import ctypes as C

import numpy as np


def oracle_branch_subset(oracle, g, sol, membuf, branches, scale=1e-4, max_auglag=50, mu_max=1e8, major_iter=2, want_x=False):
    """sol: dict of reference-layout vectors (u_curr, v_curr, z_curr, l_curr, rho); membuf: (nline, 31)
    rows as the reference's columns.  Returns (u_rows [len(branches) x 8], auglag_it, minor_it, membuf rows
    24..26 after the solve); want_x: also the solver's final point x [len(branches) x 6] and its bounds xl, xu (which variables
    ended ON a bound).  Mirrors acopf_auglag_linelimit_kernel_cpu.jl:21-168 through the oracle."""
    L = oracle.lib()
    G_ = int(g["ngen"])
    Y = np.stack([np.asarray(g[k]) for k in ("YffR", "YffI", "YftR", "YftI", "YttR", "YttI", "YtfR", "YtfI")], 1)
    n = len(branches)
    u_out = np.zeros((n, 8))
    a_ref = np.zeros(n, dtype=np.int64)
    m_ref = np.zeros(n, dtype=np.int64)
    mult = np.zeros((n, 3))
    xs, xls, xus = np.zeros((n, 6)), np.zeros((n, 6)), np.zeros((n, 6))
    for i, b in enumerate(branches):
        b = int(b)
        p0 = 2 * G_ + 8 * b
        mb = membuf[b]
        xl = np.array([g["FrVmBound"][2 * b], g["ToVmBound"][2 * b], g["FrVaBound"][2 * b], g["ToVaBound"][2 * b], -mb[28], -mb[28]])
        xu = np.array([g["FrVmBound"][2 * b + 1], g["ToVmBound"][2 * b + 1], g["FrVaBound"][2 * b + 1], g["ToVaBound"][2 * b + 1], 0.0, 0.0])
        ub_ = sol["u_curr"][p0:p0 + 8]
        x = np.array([np.sqrt(ub_[4]), np.sqrt(ub_[5]), ub_[6], ub_[7], -(ub_[0] ** 2 + ub_[1] ** 2), -(ub_[2] ** 2 + ub_[3] ** 2)])
        x = np.minimum(xu, np.maximum(xl, x))
        prm = np.zeros(31)
        prm[0:8] = sol["l_curr"][p0:p0 + 8]
        prm[8:16] = sol["rho"][p0:p0 + 8]
        prm[16:24] = sol["v_curr"][p0:p0 + 8] - sol["z_curr"][p0:p0 + 8]
        prm[24:27] = mb[24:27]
        if major_iter == 1:
            prm[26] = 10.0
        yb = np.ascontiguousarray(Y[b])
        it_, mi_ = C.c_int(), C.c_int()
        L.oracle_branch_solve(oracle._dp(x), oracle._dp(xl), oracle._dp(xu), oracle._dp(prm), oracle._dp(yb), scale, max_auglag, mu_max,
                              C.cast(C.byref(it_), oracle.c_int_p), C.cast(C.byref(mi_), oracle.c_int_p))
        a_ref[i], m_ref[i] = it_.value, mi_.value
        mult[i] = prm[24:27]
        xs[i], xls[i], xus[i] = x, xl, xu
        cs, sn = np.cos(x[2] - x[3]), np.sin(x[2] - x[3])
        vv = x[0] * x[1]
        u_out[i] = [yb[0] * x[0] ** 2 + yb[2] * vv * cs + yb[3] * vv * sn,
                    -yb[1] * x[0] ** 2 - yb[3] * vv * cs + yb[2] * vv * sn,
                    yb[4] * x[1] ** 2 + yb[6] * vv * cs - yb[7] * vv * sn,
                    -yb[5] * x[1] ** 2 - yb[7] * vv * cs - yb[6] * vv * sn,
                    x[0] ** 2, x[1] ** 2, x[2], x[3]]
    if want_x:
        return u_out, a_ref, m_ref, mult, xs, xls, xus
    return u_out, a_ref, m_ref, mult
