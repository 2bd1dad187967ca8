#!/usr/bin/env python3
"""CPU tool (no GPU): how many trips do the inner searches of a TRON model step make on the branch problems of a
steady-state ADMM iterate?  The lane code (tron_device.h / branch_solver.h) is compiled for the host with
-DEXA_HOST_COUNTERS (tests/hostcheck/hostcheck.cpp) and run on every branch of the state the oracle reaches after
N iterations of a 13 659-bus-sized grid.  Tells which loops are worth shortening on the straggler chain."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import exaadmm_loader
from oracle import oracle

CSRC = os.path.join(ROOT, "exaadmm.jl_amd", "csrc")
HC = os.path.join(ROOT, "tests", "hostcheck")
dp = C.POINTER(C.c_double)
NAMES = ["cauchy calls", "cauchy: interpolating", "cauchy trips inside region (model evaluated)", "cauchy trips outside region",
         "prsrch calls", "prsrch trips", "faces (icfs + trpcg)", "cg iterations", "icfs scaled/shifted path", "cauchy: first step outside region"]


def main():
    niter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    so = os.path.join(HC, "libhostcheck_count.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-DEXA_HOST_COUNTERS", "-I", CSRC,
                           os.path.join(HC, "hostcheck.cpp"), "-o", so, "-lm"])
    Lh = C.CDLL(so)
    ip = C.POINTER(C.c_int)
    Lh.hostcheck_branch_solve.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_int, C.c_double, ip, ip, ip]
    cnt = (C.c_longlong * 16).in_dll(Lh, "exa_cnt")
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like("case13659pegase", seed=0)
    ref = oracle.OracleModel(g, 5e1, 5e3, nthreads=min(8, os.cpu_count() or 1))
    for it in range(1, niter + 1):
        ref.inner_prestep(); ref.update_x(it); ref.update_xbar(); ref.update_z(); ref.update_l(); ref.update_residual()
    s = ref.sol
    G_, L_ = g["ngen"], g["nline"]
    Y = np.stack([g[k] for k in ("YffR", "YffI", "YftR", "YftI", "YttR", "YttI", "YtfR", "YtfI")], 1).copy()
    P = lambda a: a.ctypes.data_as(dp)
    rows = np.zeros((L_, 16), dtype=np.int64)
    ticks = np.zeros(L_, dtype=np.int64)
    minor = np.zeros(L_, dtype=np.int64)
    for b in range(L_):
        p0 = 2 * G_ + 8 * b
        mb = ref.membuf[b]
        xl = np.array([g["FrVmBound"][2 * b], g["ToVmBound"][2 * b], g["FrVaBound"][2 * b], g["ToVaBound"][2 * b], -mb[28], -mb[28]])
        xu = np.array([g["FrVmBound"][2 * b + 1], g["ToVmBound"][2 * b + 1], g["FrVaBound"][2 * b + 1], g["ToVaBound"][2 * b + 1], 0.0, 0.0])
        ub = s["u_curr"][p0:p0 + 8]
        x0 = np.array([np.sqrt(ub[4]), np.sqrt(ub[5]), ub[6], ub[7], -(ub[0] ** 2 + ub[1] ** 2), -(ub[2] ** 2 + ub[3] ** 2)])
        x0 = np.minimum(xu, np.maximum(xl, x0))
        prm = np.zeros(31)
        prm[0:8] = s["l_curr"][p0:p0 + 8]
        prm[8:16] = s["rho"][p0:p0 + 8]
        prm[16:24] = s["v_curr"][p0:p0 + 8] - s["z_curr"][p0:p0 + 8]
        prm[24:27] = mb[24:27]
        for k in range(16):
            cnt[k] = 0
        a, m, t = C.c_int(), C.c_int(), C.c_int()
        Lh.hostcheck_branch_solve(P(x0), P(xl), P(xu), P(prm), P(Y[b]), 1e-4, 50, 1e8, C.byref(a), C.byref(m), C.byref(t))
        rows[b] = list(cnt)
        ticks[b], minor[b] = t.value, m.value
    print("state: oracle iterate after %d inner iterations, %d branch problems (x-update of iteration %d)" % (niter, L_, niter + 1))
    print("solver-loop trips per branch: " + ", ".join("%d:%d" % (k, (ticks == k).sum()) for k in range(1, ticks.max() + 1)))
    for lo, hi, nm in ((1, 1, "branches with 1 trip"), (2, 2, "2 trips"), (3, 99, ">= 3 trips (the stragglers)")):
        sel = (ticks >= lo) & (ticks <= hi)
        if not sel.any():
            continue
        r = rows[sel].astype(float)
        steps = r[:, 0]                      # model steps = cauchy calls
        print("%s (n = %d): per MODEL STEP" % (nm, sel.sum()))
        for k in range(1, 10):
            print("    %-50s %.2f" % (NAMES[k], r[:, k].sum() / steps.sum()))
        tr = r[:, 2] + r[:, 3]
        per = tr / np.maximum(steps, 1)
        print("    cauchy search trips per model step, distribution over branches: " +
              ", ".join("%d:%d" % (k, ((per >= k) & (per < k + 1)).sum()) for k in range(0, int(per.max()) + 1)))


if __name__ == "__main__":
    main()
