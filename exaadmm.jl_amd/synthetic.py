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


def synthetic_grid(ngen, nline, nbus, seed=0, tight_factor=0.99, locality=64):
    """Random connected grid: spanning tree over the buses (each bus attaches to one of the
    `locality` previous buses, which gives the banded structure of real networks) plus
    extra local edges up to nline.  Returns the dict layout used by AdmmEnv(grid=...),
    zero-based indices."""
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
    bc = rng.uniform(0.0, 0.3, nline)
    Ys = 1.0 / (r + 1j * x)
    Ytt = Ys + 1j * bc / 2
    Yff, Yft, Ytf = Ytt, -Ys, -Ys
    rate = np.where(rng.random(nline) < 0.3, 0.0, rng.uniform(100.0, 900.0, nline))

    Vmin, Vmax = np.full(nbus, 0.9), np.full(nbus, 1.1)
    has_load = rng.random(nbus) < 0.6
    Pd = np.where(has_load, rng.uniform(0.0, 100.0, nbus), 0.0)
    Qd = np.where(has_load, rng.uniform(0.0, 30.0, nbus), 0.0)
    gbus = np.sort(rng.integers(0, nbus, ngen))
    Pmax = rng.uniform(50.0, 500.0, ngen)
    need = 1.3 * Pd.sum()
    if Pmax.sum() < need:
        Pmax *= need / Pmax.sum()
    g = {"ngen": ngen, "nline": nline, "nbus": nbus, "baseMVA": M, "bus_ref": 0}
    g["pgmin"] = np.zeros(ngen)
    g["pgmax"] = Pmax / M
    g["qgmin"] = -0.5 * Pmax / M
    g["qgmax"] = 0.5 * Pmax / M
    g["c2"] = rng.uniform(0.0, 0.1, ngen)
    g["c1"] = rng.uniform(1.0, 40.0, ngen)
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
    g["YshR"], g["YshI"] = np.zeros(nbus), np.zeros(nbus)

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


def sized_like(case_name, seed=0, **kw):
    ngen, nline, nbus = CASE_SIZES[case_name]
    return synthetic_grid(ngen, nline, nbus, seed=seed, **kw)
