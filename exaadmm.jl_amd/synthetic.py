"""Seeded synthetic transmission grids with the sizes of the benchmark cases.

The real MATPOWER files of BASELINE.json's configs (case1354pegase, case13659pegase,
case_ACTIVSg70k) are not available offline; bench.py looks for them in $MATPOWER_DATA or
./data and otherwise runs a size-matched grid from this generator, labelled "synthetic-..."
(never presented as the real case).  Recipe: SURVEY.md §8(d).
"""
import numpy as np

# (ngen, nline, nbus) of the named cases after the loader drops inactive components
CASE_SIZES = {
    "case9": (3, 9, 9),
    "case118": (54, 186, 118),
    "case1354pegase": (260, 1991, 1354),
    "case13659pegase": (4092, 20467, 13659),
    "case_ACTIVSg70k": (10390, 88207, 70000),
    "synthetic1M": (50000, 1000000, 500000),
}


def synthetic_grid(ngen, nline, nbus, seed=0, tight_factor=0.99, locality=64, feasible=True, recipe="acflow", cost="pegase"):
    """Random connected grid: spanning tree over the buses (each bus attaches to one of the
    `locality` previous buses, which gives the banded structure of real networks) plus
    extra local edges up to nline.  Returns the dict layout used by AdmmEnv(grid=...),
    zero-based indices.

    feasible=True (default) makes the instance a solvable OPF like the real cases instead of
    random numbers: every load is supplied by the electrically nearest generator (capacities
    are raised where needed), a DC power flow of that dispatch gives the line flows, and the
    thermal limits are set with margin above those flows.

    recipe="acflow" (default since round 2) goes one step further so that the two-level solve of the
    reference CONVERGES on the instance like it does on the real cases: a voltage profile (|V| near 1,
    angles of the DC flow) is pushed through the AC branch equations, and loads / dispatch are defined
    from the resulting bus injections, so a strictly feasible AC operating point exists by construction
    (voltages, generator P and Q, thermal limits all with margin).  recipe="dc" keeps the round-1
    instance (DC-feasible only; the ADMM diverges on it from beta ~ 2e5).
    cost="pegase": linear unit cost as in the MATPOWER pegase files (the README's rho_pq = 1e1 / 5e1
    are tuned to that cost scale); cost="market": quadratic costs c1 ~ U(10, 40), c2 ~ U(0, 0.02)
    (ACTIVSg-like, solved with rho_pq = 3e4 upstream)."""
    assert nline >= nbus - 1 and ngen >= 1
    rng = np.random.default_rng(seed)
    M = 100.0
    fr = np.empty(nline, dtype=np.int64)
    to = np.empty(nline, dtype=np.int64)
    child = np.arange(1, nbus)
    parent = child - 1 - rng.integers(0, np.minimum(child, locality))
    fr[:nbus - 1], to[:nbus - 1] = parent, child
    extra = nline - (nbus - 1)
    a = rng.integers(0, nbus, extra)
    off = rng.integers(1, max(2, locality), extra)
    b = np.minimum(nbus - 1, a + off)
    b = np.where(b == a, np.maximum(a - 1, 0), b)
    fr[nbus - 1:], to[nbus - 1:] = a, b
    perm = rng.permutation(nline)
    fr, to = fr[perm], to[perm]
    flip = rng.random(nline) < 0.5
    fr, to = np.where(flip, to, fr), np.where(flip, fr, to)

    r = rng.uniform(1e-3, 3e-2, nline)
    x = rng.uniform(1e-2, 2e-1, nline)
    bc = rng.uniform(0.0, 0.3 if not feasible else 0.06, nline)
    # ~12 % of the branches are transformers (off-nominal tap, no line charging), a few of them
    # phase shifters -- as in the pegase / ACTIVSg cases; admittances as utils/opfdata.jl:428-447
    is_tr = rng.random(nline) < 0.12
    ratio = np.where(is_tr, rng.uniform(0.94, 1.06, nline), 1.0)
    shift = np.where(is_tr & (rng.random(nline) < 0.1), rng.uniform(-5.0, 5.0, nline), 0.0)
    bc = np.where(is_tr, 0.0, bc)
    tap = ratio * np.exp(1j * shift * np.pi / 180.0)
    Ys = 1.0 / (r + 1j * x)
    Ytt = Ys + 1j * bc / 2
    Yff = Ytt / (tap * np.conj(tap))
    Yft = -Ys / np.conj(tap)
    Ytf = -Ys / tap
    unlimited = rng.random(nline) < 0.3
    rate = np.where(unlimited, 0.0, rng.uniform(100.0, 900.0, nline))

    Vmin, Vmax = np.full(nbus, 0.9), np.full(nbus, 1.1)
    has_load = rng.random(nbus) < 0.6
    Pd = np.where(has_load, rng.uniform(0.0, 100.0, nbus), 0.0)
    Qd = np.where(has_load, rng.uniform(0.0, 30.0, nbus), 0.0)
    gbus = np.sort(rng.integers(0, nbus, ngen))
    Pmax = rng.uniform(50.0, 500.0, ngen)
    need = 1.3 * Pd.sum()
    if Pmax.sum() < need:
        Pmax *= need / Pmax.sum()
    if feasible:
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        Qd = 0.25 * Pd
        # nearest generator in bus-index space (the grid is banded in that order)
        k = np.clip(np.searchsorted(gbus, np.arange(nbus)), 1, ngen - 1) if ngen > 1 else np.zeros(nbus, dtype=np.int64)
        if ngen > 1:
            left = np.abs(gbus[k - 1] - np.arange(nbus)) <= np.abs(gbus[k] - np.arange(nbus))
            k = np.where(left, k - 1, k)
        # round 1 ("dc") gave every unit 2 % more than its loads and let the reference bus absorb the surplus: on a
        # banded grid that surplus travels the whole band (angles of tens of radians at 13 659 buses, beyond the
        # +-2 pi angle bounds).  "acflow" balances every generator's area exactly: only local flows remain.
        Pg = (1.02 if recipe != "acflow" else 1.0) * np.bincount(k, weights=Pd, minlength=ngen)
        Pmax = np.maximum(Pmax, Pg / 0.6)
        inj = (np.bincount(gbus, weights=Pg, minlength=nbus) - Pd) / M
        inj[0] -= inj.sum()                      # slack at the reference bus
        w = 1.0 / x
        Bm = sp.coo_matrix((np.concatenate([w, w, -w, -w]), (np.concatenate([fr, to, fr, to]), np.concatenate([fr, to, to, fr]))),
                           shape=(nbus, nbus)).tocsr()
        theta = np.zeros(nbus)
        theta[1:] = spla.spsolve(Bm[1:, 1:].tocsc(), inj[1:])
        flow = np.abs((theta[fr] - theta[to]) * w) * M                     # MW
        rate = np.where(unlimited, 0.0, np.maximum(1.6 * flow, 60.0) * rng.uniform(1.0, 1.5, nline))
    YshR = YshI = None
    if feasible and recipe == "acflow":
        rx = rng.uniform(0.08, 0.35, nline)        # transmission-like r / x

        def dc_angles(xx):
            ww = 1.0 / (xx * (1.0 + rx * rx))      # series susceptance x / (r^2 + x^2): the DC flows then match the AC ones to first order
            Bm_ = sp.coo_matrix((np.concatenate([ww, ww, -ww, -ww]), (np.concatenate([fr, to, fr, to]), np.concatenate([fr, to, to, fr]))),
                                shape=(nbus, nbus)).tocsr()
            th = np.zeros(nbus)
            th[1:] = spla.spsolve(Bm_[1:, 1:].tocsc(), inj[1:])
            return th

        # (1) lines are as strong as what they carry, like a planned network: where the DC flow of the local dispatch
        #     opens an angle of more than 0.08 rad, the reactance shrinks (a second circuit / a higher voltage level);
        #     resistance follows with a transmission-like r/x
        theta = dc_angles(x)
        for _ in range(12):
            dth = np.abs(theta[fr] - theta[to])
            if dth.max() <= 0.10:
                break
            x = np.where(dth > 0.08, x * 0.08 / np.maximum(dth, 1e-12), x)
            theta = dc_angles(x)
        r = x * rx
        ratio = np.where(is_tr, 1.0 + (ratio - 1.0) * np.minimum(0.5, 2.0 * x), 1.0)   # taps within +-3 %, milder on strong branches
        shift = shift * np.minimum(1.0, x / 0.2)                          # phase shifters: the shift they force is b * phi; keep it ~0.2 p.u.
        tap = ratio * np.exp(1j * shift * np.pi / 180.0)
        bc = np.where(is_tr, 0.0, np.minimum(bc, 0.3 * x))               # charging in proportion to the line length
        Ys = 1.0 / (r + 1j * x)
        Ytt = Ys + 1j * bc / 2
        Yff = Ytt / (tap * np.conj(tap))
        Yft = -Ys / np.conj(tap)
        Ytf = -Ys / tap
        if np.abs(theta).max() >= 3.0:
            # Several large areas in one band (N x 70 000 buses): the angle drifts along the band until it nears the +-2 pi
            # bounds of the ACOPF.  Loads and dispatch are DERIVED from the voltage profile below (step 3), so any profile is
            # consistent: remove the long-range trend (moving average over 4 096 buses of the band) and keep the local
            # structure -- every area then looks like a one-area grid.  (Grids that stay below 3 rad -- every single-area
            # case, N areas of case13659pegase -- are not touched: their numbers are unchanged.)
            from scipy.ndimage import uniform_filter1d
            theta = theta - uniform_filter1d(theta, size=4096, mode="nearest")
            theta = theta - theta[0]
        assert np.abs(theta).max() < 3.0 and np.abs(theta[fr] - theta[to]).max() < 0.2, "synthetic grid: DC angles out of range"
        # (2) voltage magnitudes: smooth over the graph (neighbours differ by ~1e-3, not by independent draws)
        w = 1.0 / x
        Lap = sp.coo_matrix((np.concatenate([w, w, -w, -w]), (np.concatenate([fr, to, fr, to]), np.concatenate([fr, to, to, fr]))),
                            shape=(nbus, nbus)).tocsc()
        sm = spla.spsolve(sp.identity(nbus, format="csc") + Lap / np.median(w), rng.standard_normal(nbus))
        Vm = np.clip(1.0 + 0.01 * sm / max(np.abs(sm).max(), 1e-12), 0.97, 1.03)
        # bus shunts (they enter the balance): MW / MVAr at 1 p.u. (opfdata.jl:453-456)
        has_sh = rng.random(nbus) < 0.08
        YshR = np.where(has_sh, rng.uniform(0.0, 5.0, nbus), 0.0) / M
        YshI = np.where(has_sh, rng.uniform(-20.0, 20.0, nbus), 0.0) / M
        # (3) push the voltage profile through the AC branch equations; loads and dispatch follow from the injections
        V = Vm * np.exp(1j * theta)
        Sf = V[fr] * np.conj(Yff * V[fr] + Yft * V[to])      # from-end flow, p.u. (opfdata.jl:428-447)
        St = V[to] * np.conj(Ytt * V[to] + Ytf * V[fr])
        # balance of the reference's bus update (acopf_bus_kernel_cpu.jl): sum pg - Pd = sum pij + YshR w,
        # sum qg - Qd = sum qij - YshI w
        need = (np.bincount(fr, weights=Sf.real, minlength=nbus) + np.bincount(to, weights=St.real, minlength=nbus) + YshR * Vm ** 2) * M
        needq = (np.bincount(fr, weights=Sf.imag, minlength=nbus) + np.bincount(to, weights=St.imag, minlength=nbus) - YshI * Vm ** 2) * M
        ng_at = np.bincount(gbus, minlength=nbus)
        Qd = 0.25 * Pd
        # generator buses: dispatch = injection + load, shared by the units at the bus; what does not fit into
        # [10 %, 70 %] of the units' capacity (reactive: +-35 %) stays with the load
        cap = np.bincount(gbus, weights=Pmax, minlength=nbus)
        want = need + Pd
        pg_bus = np.where(ng_at > 0, np.clip(want, 0.10 * cap, 0.70 * cap), 0.0)
        Pd = pg_bus - need
        wantq = needq + Qd
        qg_bus = np.where(ng_at > 0, np.clip(wantq, -0.35 * cap, 0.35 * cap), 0.0)
        Qd = qg_bus - needq
        flowS = np.maximum(np.abs(Sf), np.abs(St)) * M
        rate = np.where(unlimited, 0.0, np.maximum(1.5 * flowS, 40.0) * rng.uniform(1.0, 1.5, nline))
        share = Pmax / np.maximum(cap[gbus], 1e-300)          # the units at a bus share its dispatch in proportion to their size
        feasible_point = {"Vm": Vm, "Va": theta, "pg": pg_bus[gbus] * share / M, "qg": qg_bus[gbus] * share / M}
    else:
        feasible_point = None
    g = {"ngen": ngen, "nline": nline, "nbus": nbus, "baseMVA": M, "bus_ref": 0}
    if feasible_point is not None:
        g["feasible_point"] = feasible_point   # (tests: the operating point the instance was built round; not read by any solver)
    g["pgmin"] = np.zeros(ngen)
    g["pgmax"] = Pmax / M
    g["qgmin"] = -0.5 * Pmax / M
    g["qgmax"] = 0.5 * Pmax / M
    c2_r, c1_r = rng.uniform(0.0, 0.1, ngen), rng.uniform(1.0, 40.0, ngen)   # (drawn in every recipe: keeps the stream aligned)
    if recipe == "acflow" and cost == "pegase":
        g["c2"], g["c1"] = np.zeros(ngen), np.ones(ngen)
    elif recipe == "acflow":
        g["c2"], g["c1"] = 0.2 * c2_r, 10.0 + 0.75 * c1_r
    else:
        g["c2"], g["c1"] = c2_r, c1_r
    g["c0"] = np.zeros(ngen)
    for nm, Y in (("Yff", Yff), ("Yft", Yft), ("Ytt", Ytt), ("Ytf", Ytf)):
        g[nm + "R"] = np.ascontiguousarray(Y.real)
        g[nm + "I"] = np.ascontiguousarray(Y.imag)
    g["FrVmBound"] = np.stack([Vmin[fr], Vmax[fr]], 1).ravel()
    g["ToVmBound"] = np.stack([Vmin[to], Vmax[to]], 1).ravel()
    fa = np.tile(np.array([-2 * np.pi, 2 * np.pi]), (nline, 1))
    ta = fa.copy()
    fa[fr == 0] = 0.0
    ta[to == 0] = 0.0
    g["FrVaBound"], g["ToVaBound"] = fa.ravel(), ta.ravel()
    g["rateA"] = np.where(rate == 0.0, 1e3, tight_factor * (rate / M) ** 2)
    g["brBusIdx"] = np.stack([fr, to], 1).ravel()
    g["Pd"], g["Qd"], g["Vmin"], g["Vmax"] = Pd, Qd, Vmin, Vmax
    # bus shunts on ~8 % of the buses (MW / MVAr at 1 p.u., divided by baseMVA: opfdata.jl:453-456)
    if YshR is None:
        has_sh = rng.random(nbus) < 0.08
        YshR = np.where(has_sh, rng.uniform(0.0, 5.0, nbus), 0.0) / M
        YshI = np.where(has_sh, rng.uniform(-20.0, 20.0, nbus), 0.0) / M
    g["YshR"], g["YshI"] = YshR, YshI

    def csr(owner, n):
        order = np.argsort(owner, kind="stable").astype(np.int64)
        start = np.zeros(n + 1, dtype=np.int64)
        np.add.at(start, owner + 1, 1)
        return np.cumsum(start), order

    g["FrStart"], g["FrIdx"] = csr(fr, nbus)
    g["ToStart"], g["ToIdx"] = csr(to, nbus)
    g["GenStart"], g["GenIdx"] = csr(gbus, nbus)
    g["gen_bus"] = gbus
    return g


# generator options that belong to a named case (the README's rho for ACTIVSg70k, 3e4 / 3e5, is tuned to market-scale costs)
CASE_KW = {"case_ACTIVSg70k": {"cost": "market"}}


def sized_like(case_name, seed=0, areas=1, **kw):
    """size-matched stand-in of a named case; areas = N: N areas of that size in one banded grid (weak scaling)"""
    ngen, nline, nbus = CASE_SIZES[case_name]
    opts = dict(CASE_KW.get(case_name, {}))
    opts.update(kw)
    return synthetic_grid(areas * ngen, areas * nline, areas * nbus, seed=seed, **opts)
