"""ORACLE -- test infrastructure only.

ctypes binding of oracle/libexaadmm_oracle.so (the C restatement of the
reference CPU path) plus an independent numpy restatement of the reference's
MATPOWER ingest.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (exaadmm.jl_amd/) never does.

Reference files followed (relative to /root/reference/src):
  utils/parse_matpower.jl:46-241   read_raw_data / add_* (status filter, per-unit)
  utils/opfdata.jl:132-254         opf_loaddata_matpower (Inf -> +-999.99)
  utils/opfdata.jl:417-456         computeAdmitances
  utils/opfdata.jl:610-618         CSR maps
  utils/opfdata.jl:694-714         bounds, rateA rule, reference-bus angle pin
  utils/opfdata.jl:817-822         brBusIdx
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int32)


def build(force=False):
    so = os.path.join(_HERE, "libexaadmm_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in
            ("tron_dense.c", "tron_dense.h", "acopf_oracle.c", "acopf_oracle.h",
             "mpacopf_oracle.c", "mpacopf_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(
            os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libexaadmm_oracle.so"])
    return so


class Grid(C.Structure):
    _fields_ = ([("ngen", C.c_int32), ("nline", C.c_int32), ("nbus", C.c_int32),
                 ("baseMVA", C.c_double)] +
                [(n, c_double_p) for n in
                 ("pgmin", "pgmax", "qgmin", "qgmax", "c2", "c1", "c0",
                  "YffR", "YffI", "YftR", "YftI", "YttR", "YttI", "YtfR", "YtfI",
                  "FrVmBound", "ToVmBound", "FrVaBound", "ToVaBound", "rateA")] +
                [("brBusIdx", c_int_p)] +
                [(n, c_double_p) for n in ("Pd", "Qd", "Vmin", "Vmax", "YshR", "YshI")] +
                [(n, c_int_p) for n in
                 ("FrStart", "FrIdx", "ToStart", "ToIdx", "GenStart", "GenIdx")])


class Params(C.Structure):
    _fields_ = [("mu_max", C.c_double), ("max_auglag", C.c_int32),
                ("initial_beta", C.c_double), ("beta", C.c_double), ("inc_c", C.c_double),
                ("theta", C.c_double), ("outer_eps", C.c_double),
                ("MAX_MULTIPLIER", C.c_double), ("outer_iterlim", C.c_int32),
                ("inner_iterlim", C.c_int32), ("scale", C.c_double), ("nthreads", C.c_int32)]


class State(C.Structure):
    _fields_ = [(n, c_double_p) for n in
                ("u_curr", "v_curr", "l_curr", "rho", "rd", "rp", "z_curr", "z_prev", "lz",
                 "Ax_plus_By", "membuf", "pgmin_curr", "pgmax_curr")]


class Info(C.Structure):
    _fields_ = [("status", C.c_int32), ("inner", C.c_int32), ("outer", C.c_int32),
                ("cumul", C.c_int32),
                ("objval", C.c_double), ("primres", C.c_double), ("dualres", C.c_double),
                ("mismatch", C.c_double), ("norm_z_curr", C.c_double),
                ("norm_z_prev", C.c_double), ("eps_pri", C.c_double),
                ("time_overall", C.c_double), ("avg_auglag_it", C.c_double),
                ("avg_minor_it", C.c_double), ("total_auglag_it", C.c_int64),
                ("total_minor_it", C.c_int64)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        gp, pp, sp, ip = C.POINTER(Grid), C.POINTER(Params), C.POINTER(State), C.POINTER(Info)
        L.oracle_default_params.argtypes = [pp]
        L.oracle_init_solution.argtypes = [gp, sp, C.c_double, C.c_double]
        L.oracle_generator_update.argtypes = [gp, sp]
        L.oracle_branch_update.argtypes = [gp, pp, sp, C.c_int, ip]
        L.oracle_bus_update.argtypes = [gp, sp]
        L.oracle_inner_prestep.argtypes = [gp, sp]
        L.oracle_update_z.argtypes = [gp, sp, C.c_double]
        L.oracle_update_l.argtypes = [gp, sp, C.c_double]
        L.oracle_update_lz.argtypes = [gp, sp, C.c_double, C.c_double]
        L.oracle_update_residual.argtypes = [gp, sp, ip]
        L.oracle_norm_z.argtypes = [gp, sp]
        L.oracle_norm_z.restype = C.c_double
        L.oracle_objective.argtypes = [gp, sp]
        L.oracle_objective.restype = C.c_double
        L.oracle_admm_two_level.argtypes = [gp, pp, sp, ip]
        L.oracle_branch_eval_f.argtypes = [c_double_p, c_double_p, C.c_double, c_double_p]
        L.oracle_branch_eval_f.restype = C.c_double
        L.oracle_branch_eval_gh.argtypes = [c_double_p, c_double_p, C.c_double, c_double_p,
                                            c_double_p, c_double_p]
        L.oracle_branch_solve.argtypes = [c_double_p, c_double_p, c_double_p, c_double_p,
                                          c_double_p, C.c_double, C.c_int, C.c_double,
                                          c_int_p, c_int_p]
        L.oracle_branch_solve.restype = C.c_int
        _LIB = L
    return _LIB


# --------------------------------------------------------------------------- #
# MATPOWER ingest (numpy restatement; independent of the product's C++ loader)
# --------------------------------------------------------------------------- #
_KEYS = ("mpc.baseMVA", "mpc.areas", "mpc.bus", "mpc.gen", "mpc.branch", "mpc.gencost")
_NCOL = {"mpc.bus": 13, "mpc.gen": 21, "mpc.branch": 13, "mpc.gencost": 7}


def read_matpower(path, case_format="matpower"):
    """parse_matpower.jl:46-88 -- section scan by first token, rows until '];',
    '%' starts a comment, fixed leading column counts."""
    with open(path) as f:
        lines = f.read().splitlines()
    ncol = dict(_NCOL)
    if case_format == "pglib":
        ncol["mpc.gen"] = 10
    out = {}
    p = 0
    while p < len(lines):
        terms = lines[p].split()
        if not terms or terms[0] not in _KEYS:
            p += 1
            continue
        key = terms[0]
        if key == "mpc.baseMVA":
            out["baseMVA"] = float(lines[p].split("=")[1].strip().replace(";", ""))
            p += 1
            continue
        p += 1
        rows = []
        while not lines[p].strip().startswith("];"):
            body = lines[p].split("%")[0].strip()
            toks = body.split()
            if key in ncol:
                rows.append([float(t.replace(";", "")) for t in toks[:ncol[key]]])
            p += 1
        if key in ncol:
            out[key] = np.array(rows, dtype=np.float64).reshape(len(rows), ncol[key])
        p += 1
    return out


def grid_from_matpower(path, tight_factor=1.0, case_format="matpower"):
    raw = read_matpower(path, case_format)
    M = raw["baseMVA"]
    bus = raw["mpc.bus"]
    gen = raw["mpc.gen"]
    br = raw["mpc.branch"]
    gc = raw["mpc.gencost"]
    nbus = bus.shape[0]
    bus_id = bus[:, 0].astype(np.int64)
    id2row = {int(b): i for i, b in enumerate(bus_id)}
    ref = np.flatnonzero(bus[:, 1].astype(np.int64) == 3)
    if len(ref) != 1:
        raise ValueError("exactly one reference bus required (opfdata.jl:158-164)")
    bus_ref = int(ref[0])

    gon = gen[:, 7].astype(np.int64) == 1          # parse_matpower.jl:129-131
    gen, gc = gen[gon], gc[gon]
    bon = br[:, 10].astype(np.int64) == 1          # parse_matpower.jl:156-158
    br = br[bon]
    ngen, nline = gen.shape[0], br.shape[0]
    if not (np.all(gc[:, 0] == 2) and np.all(gc[:, 3] == 3)):
        raise ValueError("only quadratic gencost supported (opfdata.jl:220)")

    def cap(x, v):
        return np.where(np.isinf(x), v, x)

    g = {"ngen": ngen, "nline": nline, "nbus": nbus, "baseMVA": float(M), "bus_ref": bus_ref}
    g["pgmin"] = cap(gen[:, 9] / M, -999.99)
    g["pgmax"] = cap(gen[:, 8] / M, 999.99)
    g["qgmin"] = cap(gen[:, 4] / M, -999.99)
    g["qgmax"] = cap(gen[:, 3] / M, 999.99)
    g["c2"], g["c1"], g["c0"] = gc[:, 4].copy(), gc[:, 5].copy(), gc[:, 6].copy()

    fr = np.array([id2row[int(b)] for b in br[:, 0]], dtype=np.int32)
    to = np.array([id2row[int(b)] for b in br[:, 1]], dtype=np.int32)
    r, x, b = br[:, 2], br[:, 3], br[:, 4]
    tap = np.where(br[:, 8] == 0.0, 1.0, br[:, 8]) * np.exp(1j * br[:, 9] * math.pi / 180)
    Ys = 1.0 / (r + 1j * x)
    Ytt = Ys + 1j * b / 2
    Yff = Ytt / (tap * np.conj(tap))
    Yft = -Ys / np.conj(tap)
    Ytf = -Ys / tap
    for nm, Y in (("Yff", Yff), ("Yft", Yft), ("Ytt", Ytt), ("Ytf", Ytf)):
        g[nm + "R"] = np.ascontiguousarray(Y.real)
        g[nm + "I"] = np.ascontiguousarray(Y.imag)
    Vmin, Vmax = bus[:, 12].copy(), bus[:, 11].copy()
    g["FrVmBound"] = np.stack([Vmin[fr], Vmax[fr]], 1).ravel()
    g["ToVmBound"] = np.stack([Vmin[to], Vmax[to]], 1).ravel()
    fa = np.tile(np.array([-2 * math.pi, 2 * math.pi]), (nline, 1))
    ta = fa.copy()
    fa[fr == bus_ref] = 0.0
    ta[to == bus_ref] = 0.0
    g["FrVaBound"], g["ToVaBound"] = fa.ravel(), ta.ravel()
    g["rateA"] = np.where(br[:, 5] == 0.0, 1e3, tight_factor * (br[:, 5] / M) ** 2)
    g["brBusIdx"] = np.stack([fr, to], 1).ravel().astype(np.int32)

    g["Pd"], g["Qd"] = bus[:, 2].copy(), bus[:, 3].copy()
    g["Vmin"], g["Vmax"] = Vmin, Vmax
    g["YshR"], g["YshI"] = bus[:, 4] / M, bus[:, 5] / M

    def csr(owner, n):
        order = np.argsort(owner, kind="stable").astype(np.int32)
        start = np.zeros(n + 1, dtype=np.int32)
        np.add.at(start, owner + 1, 1)
        return np.cumsum(start).astype(np.int32), order

    g["FrStart"], g["FrIdx"] = csr(fr, nbus)
    g["ToStart"], g["ToIdx"] = csr(to, nbus)
    gbus = np.array([id2row[int(b)] for b in gen[:, 0]], dtype=np.int32)
    g["GenStart"], g["GenIdx"] = csr(gbus, nbus)
    g["gen_bus"] = gbus
    for k, v in list(g.items()):
        if isinstance(v, np.ndarray) and v.dtype == np.float64:
            g[k] = np.ascontiguousarray(v)
    return g


_GRID_F64 = ("pgmin", "pgmax", "qgmin", "qgmax", "c2", "c1", "c0",
             "YffR", "YffI", "YftR", "YftI", "YttR", "YttI", "YtfR", "YtfI",
             "FrVmBound", "ToVmBound", "FrVaBound", "ToVaBound", "rateA",
             "Pd", "Qd", "Vmin", "Vmax", "YshR", "YshI")
_GRID_I32 = ("brBusIdx", "FrStart", "FrIdx", "ToStart", "ToIdx", "GenStart", "GenIdx")
_STATE = ("u_curr", "v_curr", "l_curr", "rho", "rd", "rp", "z_curr", "z_prev", "lz", "Ax_plus_By")


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


class OracleModel:
    """env + ModelAcopf of the reference rolled into one object, CPU oracle backend."""

    def __init__(self, grid, rho_pq=400.0, rho_va=40000.0, **params):
        self.L = lib()
        self.g = {k: (np.ascontiguousarray(v, dtype=np.float64) if k in _GRID_F64 else
                      np.ascontiguousarray(v, dtype=np.int32) if k in _GRID_I32 else v)
                  for k, v in grid.items()}
        g = self.g
        self.ngen, self.nline, self.nbus = int(g["ngen"]), int(g["nline"]), int(g["nbus"])
        self.nvar = 2 * self.ngen + 8 * self.nline
        self.cg = Grid()
        self.cg.ngen, self.cg.nline, self.cg.nbus = self.ngen, self.nline, self.nbus
        self.cg.baseMVA = float(g["baseMVA"])
        for k in _GRID_F64:
            setattr(self.cg, k, _dp(g[k]))
        for k in _GRID_I32:
            setattr(self.cg, k, _ip(g[k]))
        self.par = Params()
        self.L.oracle_default_params(C.byref(self.par))
        for k, v in params.items():
            setattr(self.par, k, v)
        self.sol = {k: np.zeros(self.nvar) for k in _STATE}
        self.membuf = np.zeros((self.nline, 31))   # row I = column I of the 31 x nline reference buffer
        self.pgmin_curr = np.zeros(self.ngen)
        self.pgmax_curr = np.zeros(self.ngen)
        self.cs = State()
        for k in _STATE:
            setattr(self.cs, k, _dp(self.sol[k]))
        self.cs.membuf = _dp(self.membuf)
        self.cs.pgmin_curr = _dp(self.pgmin_curr)
        self.cs.pgmax_curr = _dp(self.pgmax_curr)
        self.info = Info()
        self.L.oracle_init_solution(C.byref(self.cg), C.byref(self.cs), rho_pq, rho_va)

    # step-wise plugin surface (docs/src/dev.md:65-155)
    def inner_prestep(self):
        self.L.oracle_inner_prestep(C.byref(self.cg), C.byref(self.cs))

    def update_x(self, major_iter):
        self.L.oracle_generator_update(C.byref(self.cg), C.byref(self.cs))
        self.L.oracle_branch_update(C.byref(self.cg), C.byref(self.par), C.byref(self.cs),
                                    int(major_iter), C.byref(self.info))

    def update_x_gen(self):
        self.L.oracle_generator_update(C.byref(self.cg), C.byref(self.cs))

    def update_x_line(self, major_iter):
        self.L.oracle_branch_update(C.byref(self.cg), C.byref(self.par), C.byref(self.cs),
                                    int(major_iter), C.byref(self.info))

    def update_xbar(self):
        self.L.oracle_bus_update(C.byref(self.cg), C.byref(self.cs))

    def update_z(self):
        self.L.oracle_update_z(C.byref(self.cg), C.byref(self.cs), self.par.beta)

    def update_l(self):
        self.L.oracle_update_l(C.byref(self.cg), C.byref(self.cs), self.par.beta)

    def update_lz(self):
        self.L.oracle_update_lz(C.byref(self.cg), C.byref(self.cs), self.par.beta,
                                self.par.MAX_MULTIPLIER)

    def update_residual(self):
        self.L.oracle_update_residual(C.byref(self.cg), C.byref(self.cs), C.byref(self.info))
        return (self.info.primres, self.info.dualres, self.info.norm_z_curr, self.info.mismatch)

    def norm_z(self):
        return self.L.oracle_norm_z(C.byref(self.cg), C.byref(self.cs))

    def objective(self):
        return self.L.oracle_objective(C.byref(self.cg), C.byref(self.cs))

    def admm_two_level(self):
        self.L.oracle_admm_two_level(C.byref(self.cg), C.byref(self.par), C.byref(self.cs),
                                     C.byref(self.info))
        return self.info


def solve_acopf(case, outer_iterlim=20, inner_iterlim=1000, rho_pq=400.0, rho_va=40000.0,
                scale=1e-4, tight_factor=1.0, outer_eps=2e-4, nthreads=1, grid=None):
    """interface/solve_acopf.jl:1-46 on the oracle backend."""
    if grid is None:
        grid = grid_from_matpower(case, tight_factor=tight_factor)
    m = OracleModel(grid, rho_pq, rho_va, scale=scale, outer_eps=outer_eps,
                    outer_iterlim=outer_iterlim, inner_iterlim=inner_iterlim, nthreads=nthreads)
    m.admm_two_level()
    return m


# --------------------------------------------------------------------------- #
# Multi-period ACOPF (ModelMpacopf), oracle backend -- mpacopf_oracle.c
# --------------------------------------------------------------------------- #
_RAMP = ("u_curr", "l_curr", "rho", "rd", "rp", "z_curr", "z_prev", "lz", "Ax_plus_By", "s_curr")


class Ramp(C.Structure):
    _fields_ = [(n, c_double_p) for n in _RAMP + ("gen_membuf",)]


class MP(C.Structure):
    _fields_ = [("T", C.c_int32), ("grids", C.POINTER(Grid)), ("states", C.POINTER(State)),
                ("ramps", C.POINTER(Ramp)), ("ramp_rate", c_double_p), ("sub_info", C.POINTER(Info))]


def _mp_lib():
    L = lib()
    if not getattr(L, "_mp_ready", False):
        mp, pp, ip = C.POINTER(MP), C.POINTER(Params), C.POINTER(Info)
        L.oracle_mp_nvar.argtypes = [mp]
        L.oracle_mp_init_solution.argtypes = [mp, C.c_double, C.c_double]
        L.oracle_mp_inner_prestep.argtypes = [mp]
        L.oracle_mp_update_x.argtypes = [mp, pp, C.c_int]
        L.oracle_mp_update_xbar.argtypes = [mp]
        L.oracle_mp_update_z.argtypes = [mp, C.c_double]
        L.oracle_mp_update_l.argtypes = [mp, C.c_double]
        L.oracle_mp_update_lz.argtypes = [mp, C.c_double, C.c_double]
        L.oracle_mp_update_residual.argtypes = [mp, ip]
        L.oracle_mp_norm_z_prev.argtypes = [mp]
        L.oracle_mp_norm_z_prev.restype = C.c_double
        L.oracle_mp_objective.argtypes = [mp]
        L.oracle_mp_objective.restype = C.c_double
        L.oracle_mp_admm_two_level.argtypes = [mp, pp, ip]
        L.oracle_genramp_eval_f.restype = C.c_double
        L.oracle_genramp_eval_f.argtypes = [c_double_p, c_double_p] + [C.c_double] * 5
        L.oracle_genramp_eval_gh.argtypes = [c_double_p, c_double_p] + [C.c_double] * 5 + [c_double_p, c_double_p]
        L.oracle_genramp_solve.restype = C.c_int
        L.oracle_genramp_solve.argtypes = ([c_double_p] * 4 + [C.c_double] * 5 + [C.c_int, C.c_double, c_int_p, c_int_p])
        L._mp_ready = True
    return L


class OracleMpModel:
    """env + ModelMpacopf of the reference (mpacopf_model.jl:40-112), CPU oracle backend.
    `pd`, `qd`: (nbus, T) load profiles in MW / MVAr (columns = periods), as get_load returns them."""

    def __init__(self, grid, pd, qd, rho_pq=400.0, rho_va=40000.0, ramp_ratio=0.02, **params):
        self.L = _mp_lib()
        pd = np.asarray(pd, dtype=np.float64)
        qd = np.asarray(qd, dtype=np.float64)
        self.T = T = pd.shape[1]
        self.periods = []
        for t in range(T):
            gt = dict(grid)
            gt["Pd"] = np.ascontiguousarray(pd[:, t])
            gt["Qd"] = np.ascontiguousarray(qd[:, t])
            self.periods.append(OracleModel(gt, rho_pq, rho_va, **params))
        m0 = self.periods[0]
        self.ngen, self.nline, self.nbus = m0.ngen, m0.nline, m0.nbus
        self.par = m0.par
        G = self.ngen
        self.ramp = [{k: np.zeros(G) for k in _RAMP} for _ in range(T)]
        self.gen_membuf = [np.zeros((G, 8)) for _ in range(T)]
        self.ramp_rate = np.ascontiguousarray(ramp_ratio * m0.g["pgmax"])
        self._grids = (Grid * T)(*[p.cg for p in self.periods])
        self._states = (State * T)(*[p.cs for p in self.periods])
        self._ramps = (Ramp * T)()
        for t in range(T):
            for k in _RAMP:
                setattr(self._ramps[t], k, _dp(self.ramp[t][k]))
            self._ramps[t].gen_membuf = _dp(self.gen_membuf[t])
        self._sub = (Info * T)()
        self.mp = MP(T, self._grids, self._states, self._ramps, _dp(self.ramp_rate), self._sub)
        self.info = Info()
        self.nvar = self.L.oracle_mp_nvar(C.byref(self.mp))
        self.L.oracle_mp_init_solution(C.byref(self.mp), rho_pq, rho_va)

    def inner_prestep(self):
        self.L.oracle_mp_inner_prestep(C.byref(self.mp))

    def update_x(self, major_iter):
        self.L.oracle_mp_update_x(C.byref(self.mp), C.byref(self.par), int(major_iter))

    def update_xbar(self):
        self.L.oracle_mp_update_xbar(C.byref(self.mp))

    def update_z(self):
        self.L.oracle_mp_update_z(C.byref(self.mp), self.par.beta)

    def update_l(self):
        self.L.oracle_mp_update_l(C.byref(self.mp), self.par.beta)

    def update_lz(self):
        self.L.oracle_mp_update_lz(C.byref(self.mp), self.par.beta, self.par.MAX_MULTIPLIER)

    def update_residual(self):
        self.L.oracle_mp_update_residual(C.byref(self.mp), C.byref(self.info))
        return (self.info.primres, self.info.dualres, self.info.norm_z_curr, self.info.mismatch)

    def norm_z_prev(self):
        return self.L.oracle_mp_norm_z_prev(C.byref(self.mp))

    def objective(self):
        return self.L.oracle_mp_objective(C.byref(self.mp))

    def admm_two_level(self):
        self.L.oracle_mp_admm_two_level(C.byref(self.mp), C.byref(self.par), C.byref(self.info))
        return self.info


def solve_mpacopf(grid, pd, qd, outer_iterlim=20, inner_iterlim=1000, rho_pq=400.0, rho_va=40000.0,
                  scale=1e-4, outer_eps=2e-4, ramp_ratio=0.02, warm_start=True):
    """interface/solve_mpacopf.jl:1-37 on the oracle backend."""
    m = OracleMpModel(grid, pd, qd, rho_pq, rho_va, ramp_ratio=ramp_ratio, scale=scale, outer_eps=outer_eps,
                      outer_iterlim=outer_iterlim, inner_iterlim=inner_iterlim)
    if warm_start:
        for p in m.periods:          # each period alone, then re-initialise the ramp blocks (:26-31)
            p.par = m.par
            p.admm_two_level()
        m.L.oracle_mp_init_solution(C.byref(m.mp), rho_pq, rho_va)
    m.admm_two_level()
    return m
