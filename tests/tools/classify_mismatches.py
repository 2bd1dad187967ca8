#!/usr/bin/env python3
"""CPU tool (no GPU): how far is the product's DEFAULT branch arithmetic from the CPU path, and why?

The lane code of exaadmm.jl_amd/csrc (tron_device.h / branch_solver.h) is compiled for the host twice through
tests/hostcheck/hostcheck.cpp -- default ("fast": explicit FMA algebra, reciprocal pivots, unscaled-Cholesky
shortcut, re-derived f/g/H) and -DEXA_EXACT_ORACLE_ARITH -- and both are run beside the oracle on the branch
problems of realistic ADMM states (the states of tests/test_gpu_parity.py::test_branch_kernel_vs_oracle_on_perturbed_states:
13 659-bus-sized grids, a few oracle iterations, optional noise).  Prints, per state, the exact number of branches whose
(augmented-Lagrangian, TRON minor) iteration counts differ from the oracle's, the class of every difference and the
distance of the solutions.  Output committed as profiles/archive_r01_r02/r02_branch_arith_mismatch_classes.txt."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import exaadmm_loader
from oracle import oracle

CSRC = os.path.join(ROOT, "exaadmm.jl_amd", "csrc")
HC = os.path.join(ROOT, "tests", "hostcheck")
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


def build(name, extra):
    so = os.path.join(HC, name)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-I", CSRC,
                           os.path.join(HC, "hostcheck.cpp"), "-o", so, "-lm"] + extra)
    L = C.CDLL(so)
    L.hostcheck_branch_solve.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_int, C.c_double, ip, ip, ip]
    return L


def _p(a):
    return a.ctypes.data_as(dp)


def main():
    syn = exaadmm_loader.load_submodule("synthetic")
    fast = build("libhostcheck.so", [])
    exact = build("libhostcheck_exact.so", ["-DEXA_EXACT_ORACLE_ARITH"])
    OL = oracle.lib()
    nthreads = min(8, os.cpu_count() or 1)
    tot = {"n": 0, "fast": 0, "exact": 0}
    for seed, noise in ((0, 0.0), (1, 0.02), (2, 0.2)):
        g = syn.sized_like("case13659pegase", seed=seed)
        rho = (5e1, 5e3)
        ref = oracle.OracleModel(g, *rho, nthreads=nthreads)
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
        s = ref.sol
        Y = np.stack([g[k] for k in ("YffR", "YffI", "YftR", "YftI", "YttR", "YttI", "YtfR", "YtfI")], 1).copy()
        classes = {}
        nf = ne = 0
        dx_same = dx_diff = 0.0
        bitexact = 0
        for b in range(L_):
            p0 = 2 * G_ + 8 * b
            mb = ref.membuf[b]
            xl = np.array([g["FrVmBound"][2 * b], g["ToVmBound"][2 * b], g["FrVaBound"][2 * b], g["ToVaBound"][2 * b], -mb[28], -mb[28]])
            xu = np.array([g["FrVmBound"][2 * b + 1], g["ToVmBound"][2 * b + 1], g["FrVaBound"][2 * b + 1], g["ToVaBound"][2 * b + 1], 0.0, 0.0])
            ub_ = s["u_curr"][p0:p0 + 8]
            x0 = np.array([np.sqrt(ub_[4]), np.sqrt(ub_[5]), ub_[6], ub_[7], -(ub_[0] ** 2 + ub_[1] ** 2), -(ub_[2] ** 2 + ub_[3] ** 2)])
            x0 = np.minimum(xu, np.maximum(xl, x0))
            prm0 = np.zeros(31)
            prm0[0:8] = s["l_curr"][p0:p0 + 8]
            prm0[8:16] = s["rho"][p0:p0 + 8]
            prm0[16:24] = s["v_curr"][p0:p0 + 8] - s["z_curr"][p0:p0 + 8]
            prm0[24:27] = mb[24:27]
            out = {}
            xo, po = x0.copy(), prm0.copy()
            a, m = C.c_int(), C.c_int()
            OL.oracle_branch_solve(_p(xo), _p(xl), _p(xu), _p(po), _p(Y[b]), 1e-4, 50, 1e8,
                                   C.cast(C.byref(a), oracle.c_int_p), C.cast(C.byref(m), oracle.c_int_p))
            out["oracle"] = (a.value, m.value, xo, po)
            for nm, Lh in (("fast", fast), ("exact", exact)):
                xh, ph = x0.copy(), prm0.copy()
                a, m, t = C.c_int(), C.c_int(), C.c_int()
                Lh.hostcheck_branch_solve(_p(xh), _p(xl), _p(xu), _p(ph), _p(Y[b]), 1e-4, 50, 1e8, C.byref(a), C.byref(m), C.byref(t))
                out[nm] = (a.value, m.value, xh, ph)
            ao, mo, xo, po = out["oracle"]
            af, mf, xf, pf = out["fast"]
            ae, me, xe, pe = out["exact"]
            bitexact += np.array_equal(xe, xo) and np.array_equal(pe[24:27], po[24:27]) and (ae, me) == (ao, mo)
            d = np.abs(xf - xo).max()
            if (af, mf) != (ao, mo):
                nf += 1
                if af != ao:
                    key = "augmented-Lagrangian iterations differ (%+d)" % (af - ao)
                else:
                    key = "same auglag iterations, TRON minor iterations differ (%+d)" % (mf - mo)
                if pf[26] != po[26]:
                    key += ", final penalty differs"
                classes[key] = classes.get(key, 0) + 1
                dx_diff = max(dx_diff, d)
            else:
                dx_same = max(dx_same, d)
            ne += (ae, me) != (ao, mo)
        print("state seed=%d noise=%.2f: %d branch problems" % (seed, noise, L_))
        print("  exact build : %d count mismatches; %d / %d bitwise identical to the oracle (x, multipliers, penalty, counts)" % (ne, bitexact, L_))
        print("  fast build  : %d count mismatches (%.4f %%); max |dx| where counts agree %.2e, where they differ %.2e" % (nf, 100.0 * nf / L_, dx_same, dx_diff))
        for k in sorted(classes, key=lambda q: -classes[q]):
            print("      %5d  %s" % (classes[k], k))
        tot["n"] += L_; tot["fast"] += nf; tot["exact"] += ne
    print("TOTAL: %d problems; exact build %d mismatches; fast build %d mismatches (%.4f %%)" % (tot["n"], tot["exact"], tot["fast"], 100.0 * tot["fast"] / tot["n"]))


if __name__ == "__main__":
    main()
