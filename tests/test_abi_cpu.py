"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/exaadmm_hip.h declares, the native MATPOWER loader agrees with the oracle's numpy
loader, and compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import CASE9, CASE118, ROOT
import exaadmm_loader


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__
    __graft_entry__.build()
    return exaadmm_loader.load_package()


def test_every_declared_symbol_is_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "exaadmm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(exaadmm_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 35
    lib = C.CDLL(pkg.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


@pytest.mark.parametrize("case", [CASE9, CASE118], ids=["case9", "case118"])
def test_native_matpower_loader_matches_oracle_loader(pkg, oracle, case):
    """(case118 adds what case9 lacks: transformers, bus shunts, parallel circuits, unlimited ratings,
    buses without load, 54 generators on 118 buses)"""
    a = pkg.opf_loaddata(case)
    b = oracle.grid_from_matpower(case)
    assert (a["ngen"], a["nline"], a["nbus"]) == (b["ngen"], b["nline"], b["nbus"])
    for k in pkg.Grid._F64 + pkg.Grid._BUS:
        assert np.allclose(a[k], b[k], rtol=1e-15, atol=0), k
    for k in pkg.Grid._IDX + ("brBusIdx",):
        assert np.array_equal(a[k], b[k].astype(np.int64)), k


def test_loader_filters_inactive_and_applies_tight_factor(pkg, tmp_path):
    src = open(CASE9).read()
    # switch generator 3 and branch 9 off; rateA of branch 1 -> 0 (becomes 1e3)
    src = src.replace("3\t85\t-10.95\t300\t-300\t1.025\t100\t1", "3\t85\t-10.95\t300\t-300\t1.025\t100\t0")
    src = src.replace("9\t4\t0.01\t0.085\t0.176\t250\t250\t250\t0\t0\t1", "9\t4\t0.01\t0.085\t0.176\t250\t250\t250\t0\t0\t0")
    src = src.replace("1\t4\t0\t0.0576\t0\t250", "1\t4\t0\t0.0576\t0\t0")
    p = tmp_path / "c.m"
    p.write_text(src)
    g = pkg.opf_loaddata(str(p), tight_factor=0.99)
    assert (g["ngen"], g["nline"]) == (2, 8)
    assert g["rateA"][0] == 1e3
    assert np.isclose(g["rateA"][1], 0.99 * 6.25)


def test_loader_reports_missing_file(pkg):
    with pytest.raises(pkg.ExaAdmmError):
        pkg.opf_loaddata("/nonexistent/case.m")


def test_compute_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.ExaAdmmError, match="no CPU fallback|no HIP device|failed"):
        env = pkg.AdmmEnv(CASE9, 4e2, 4e4, verbose=0)
        pkg.ModelAcopf(env)


def test_product_never_imports_oracle():
    """no import / include / dlopen of anything under oracle/ from the product tree"""
    bad = re.compile(r"(import\s+oracle|from\s+oracle|oracle/|oracle\.py|libexaadmm_oracle|acopf_oracle\.h|tron_dense\.h)")
    # product tree, diagnostics, Julia binding, loader: only tests/, smoke() and bench.py's cpu_baseline may use the oracle
    for pdir in ("exaadmm.jl_amd", "scripts", "julia", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, pdir)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".jl", "Makefile")):
                    txt = open(os.path.join(dp, f)).read()
                    assert not bad.search(txt), (dp, f)
    assert not bad.search(open(os.path.join(ROOT, "exaadmm_loader.py")).read())


def test_synthetic_grid_is_connected_and_sized(pkg):
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like("case1354pegase")
    assert (g["ngen"], g["nline"], g["nbus"]) == syn.CASE_SIZES["case1354pegase"]
    # connectivity by union-find
    parent = list(range(g["nbus"]))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    bb = g["brBusIdx"].reshape(-1, 2)
    for f, t in bb:
        parent[find(f)] = find(t)
    assert len({find(i) for i in range(g["nbus"])}) == 1
    assert g["FrStart"][-1] == g["nline"] and g["GenStart"][-1] == g["ngen"]
    assert (bb[:, 0] != bb[:, 1]).all()


def test_argument_errors_are_reported_not_thrown(pkg):
    """error convention of the boundary: negative status + message, no exception / abort"""
    import ctypes as C
    L = pkg.library()
    h = C.c_void_p()
    g = pkg.Grid()          # all-zero grid: nline == 0
    p = pkg.Parameters()
    L.exaadmm_default_params(C.byref(p))
    assert (p.mu_max, p.max_auglag, p.initial_beta, p.inc_c, p.theta) == (1e8, 50, 1e3, 6.0, 0.8)
    assert (p.outer_iterlim, p.inner_iterlim, p.scale, p.MAX_MULTIPLIER) == (20, 1000, 1e-4, 1e12)
    rc = L.exaadmm_create(C.byref(g), C.byref(p), 0, C.byref(h))
    assert rc == -1 and b"empty grid" in L.exaadmm_last_error()
    assert L.exaadmm_update_x(None, 1) == -1
    assert L.exaadmm_nvar(None) == -1
    assert L.exaadmm_destroy(None) == 0
    ver = L.exaadmm_version().decode()
    assert ver.startswith("exaadmm-hip")
    # the code-generation flags travel in the version string: the branch kernel is only known to compile
    # correctly with the basic SGPR allocator and without FP contraction (csrc/Makefile pins both)
    assert "-mllvm -sgpr-regalloc=basic" in ver and "-ffp-contract=off" in ver, ver
    # (the measured build also preloads the kernel arguments into SGPRs; kernels keep an entry for firmware that does not)
    assert "-amdgpu-kernarg-preload-count=16" in ver and "src:" in ver, ver
    # multi-period entry points: same convention
    assert L.exaadmm_mp_create(C.byref(g), C.byref(p), 3, None, None, 0.02, 0, C.byref(h)) == -1
    assert b"null argument" in L.exaadmm_last_error()
    assert L.exaadmm_mp_periods(None) == -1 and L.exaadmm_mp_nvar(None) == -1
    assert L.exaadmm_mp_update_x_gen(None, 1) == -1
    assert L.exaadmm_mp_download_ramp(None, 0, None, 0) == -1


def test_multi_period_model_fails_loudly_without_gpu(pkg):
    """no CPU path for the multi-period model either"""
    import numpy as np
    env = pkg.AdmmEnv(CASE9, 4e2, 4e4, verbose=0)
    load = (np.tile(env.data["Pd"][:, None], (1, 3)), np.tile(env.data["Qd"][:, None], (1, 3)))
    with pytest.raises(pkg.ExaAdmmError, match="no HIP device|no CPU fallback"):
        pkg.ModelMpacopf(env, load, end_period=3)
    with pytest.raises(pkg.ExaAdmmError, match="out of range"):
        pkg.ModelMpacopf(env, load, start_period=2, end_period=5)


def test_dead_reference_options_are_rejected(pkg):
    with pytest.raises(pkg.ExaAdmmError, match="use_linelimit"):
        pkg.AdmmEnv(CASE9, 4e2, 4e4, use_linelimit=False)
    with pytest.raises(pkg.ExaAdmmError, match="use_projection"):
        pkg.AdmmEnv(CASE9, 4e2, 4e4, use_projection=True)


def test_loader_taps_phase_shifters_shunts_inf_limits(pkg, oracle, tmp_path):
    """admittances with off-nominal taps and phase shifts, bus shunts, Inf generator limits
    (parse_matpower.jl:137-142, opfdata.jl:199-206,428-456): native loader == numpy restatement"""
    src = open(CASE9).read()
    src = src.replace("4\t5\t0.017\t0.092\t0.158\t250\t250\t250\t0\t0\t1", "4\t5\t0.017\t0.092\t0.158\t250\t250\t250\t0.97\t0\t1")
    src = src.replace("6\t7\t0.0119\t0.1008\t0.209\t150\t150\t150\t0\t0\t1", "6\t7\t0.0119\t0.1008\t0.209\t150\t150\t150\t1.03\t-3.5\t1")
    src = src.replace("5\t1\t90\t30\t0\t0\t1", "5\t1\t90\t30\t2.5\t-19\t1")
    src = src.replace("2\t163\t6.54\t300\t-300\t1.025\t100\t1\t300\t10", "2\t163\t6.54\tInf\t-Inf\t1.025\t100\t1\tInf\t10")
    p = tmp_path / "c.m"
    p.write_text(src)
    a = pkg.opf_loaddata(str(p))
    b = oracle.grid_from_matpower(str(p))
    for k in pkg.Grid._F64 + pkg.Grid._BUS:
        assert np.allclose(a[k], b[k], rtol=1e-14, atol=0), k
    assert a["YffR"][1] != a["YttR"][1] and a["YftI"][4] != a["YtfI"][4]       # tap / phase shift took effect
    assert a["YshR"][4] == 0.025 and a["YshI"][4] == -0.19
    assert a["qgmax"][1] == 999.99 and a["qgmin"][1] == -999.99 and a["pgmax"][1] == 999.99


def test_loader_pglib_format_has_ten_generator_columns(pkg, oracle, tmp_path):
    src = open(CASE9).read()
    lines = []
    for ln in src.splitlines():
        t = ln.split("\t")
        if len(t) >= 22 and t[0] == "" and t[8] in ("1",) and t[7] == "100":   # a generator row: keep 10 columns
            ln = "\t".join(t[:11]) + ";"
        lines.append(ln)
    p = tmp_path / "pglib.m"
    p.write_text("\n".join(lines))
    a = pkg.opf_loaddata(str(p), case_format="pglib")
    b = oracle.grid_from_matpower(CASE9)
    assert a["ngen"] == 3 and np.allclose(a["pgmax"], b["pgmax"]) and np.allclose(a["c2"], b["c2"])


def test_c_driver_builds_as_plain_c_and_fails_loudly_without_gpu(pkg):
    """tools/exaadmm_solve.c: the header is valid C11, the driver links against the library, and
    without a GPU it exits with code 2 and the library's message (no CPU path)."""
    import subprocess
    csrc = os.path.join(ROOT, "exaadmm.jl_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "../../tools/exaadmm_solve"])
    exe = os.path.join(ROOT, "tools", "exaadmm_solve")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "no_such_case.m"), "--quiet"], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot open" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, CASE9, "--quiet"], capture_output=True, text=True)
        assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.parametrize("case,kw", [("case118", {}), ("case1354pegase", {}), ("case13659pegase", {}), ("case1354pegase", {"cost": "market"})])
def test_synthetic_instances_have_a_strictly_feasible_operating_point(case, kw):
    """The benchmark instances (synthetic.py, recipe "acflow", seed 0) are built round an AC operating point; check,
    with the reference's own equations (branch flows opfdata.jl:428-447 / acopf_eval_linelimit_kernel_cpu.jl:11-16, bus
    balance acopf_bus_kernel_cpu.jl), that this point satisfies every constraint of the ACOPF with margin: power balance
    at every bus, voltage / angle / generator boxes, thermal limits.  That is what makes the reference's two-level solve
    converge on them (round 1's DC-only instance diverged)."""
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like(case, seed=0, **kw)
    fp = g["feasible_point"]
    Vm, Va, pg, qg = fp["Vm"], fp["Va"], fp["pg"], fp["qg"]
    L, B, M = g["nline"], g["nbus"], g["baseMVA"]
    bb = np.asarray(g["brBusIdx"]).reshape(L, 2)
    fr, to = bb[:, 0], bb[:, 1]
    vi, vj, th = Vm[fr], Vm[to], Va[fr] - Va[to]
    cs, sn = np.cos(th), np.sin(th)
    pij = g["YffR"] * vi ** 2 + g["YftR"] * vi * vj * cs + g["YftI"] * vi * vj * sn
    qij = -g["YffI"] * vi ** 2 - g["YftI"] * vi * vj * cs + g["YftR"] * vi * vj * sn
    pji = g["YttR"] * vj ** 2 + g["YtfR"] * vi * vj * cs - g["YtfI"] * vi * vj * sn
    qji = -g["YttI"] * vj ** 2 - g["YtfI"] * vi * vj * cs - g["YtfR"] * vi * vj * sn
    gb = np.asarray(g["gen_bus"])
    pbal = np.bincount(gb, weights=pg, minlength=B) - g["Pd"] / M - np.bincount(fr, weights=pij, minlength=B) \
        - np.bincount(to, weights=pji, minlength=B) - g["YshR"] * Vm ** 2
    qbal = np.bincount(gb, weights=qg, minlength=B) - g["Qd"] / M - np.bincount(fr, weights=qij, minlength=B) \
        - np.bincount(to, weights=qji, minlength=B) + g["YshI"] * Vm ** 2
    scale = max(1.0, np.abs(pij).max())
    assert np.abs(pbal).max() <= 1e-9 * scale and np.abs(qbal).max() <= 1e-9 * scale
    assert (Vm > g["Vmin"] + 0.05).all() and (Vm < g["Vmax"] - 0.05).all()
    assert np.abs(Va).max() < np.pi                    # angle boxes are +-2 pi; the reference bus sits at 0
    assert Va[0] == 0.0
    assert (pg >= g["pgmin"] + 0.05 * g["pgmax"]).all() and (pg <= 0.75 * g["pgmax"]).all()
    assert (qg >= 0.8 * g["qgmin"]).all() and (qg <= 0.8 * g["qgmax"]).all()
    lim = np.asarray(g["rateA"])
    assert ((pij ** 2 + qij ** 2) <= 0.5 * lim).all() and ((pji ** 2 + qji ** 2) <= 0.5 * lim).all()   # flows <= 1/1.5 of the rating


def test_option_defaults_of_the_python_mirror_are_the_librarys(pkg):
    """The Python mirror sets EVERY option when a model is created (environment switches translated into options): its table of
    defaults must be the library's own (csrc/exaadmm_hip.hip struct Options), its keys must be exactly the keys of the table behind
    exaadmm_set_option in the shipped build, include/exaadmm_hip.h must document every one of them -- and there are at most fifteen
    (switches of experiments that lost are gone, test switches live in the test build's exaadmm_test_hook)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "exaadmm.jl_amd", "csrc", "exaadmm_hip.hip")).read()
    body = src[src.index("struct Options {"):]
    body = body[:body.index("};")]
    lib = {k: int(v) for k, v in re.findall(r"(\w+)\s*=\s*(-?\d+)\s*[,;]", body)}
    assert lib, "struct Options not parsed"
    tab = src[src.index("const OptKey OPT_KEYS[] = {"):]
    tab = tab[:tab.index("};")]
    shipped = re.sub(r"#if defined\(EXA_PROFILE_PHASES\).*?#endif", "", tab, flags=re.S)   # (prof_resident: the diagnostic build's only)
    table = set(re.findall(r'\{"(\w+)",\s*&Options::', shipped))
    assert table == set(pkg._OPTION_DEFAULTS), table ^ set(pkg._OPTION_DEFAULTS)
    assert len(table) <= 15, sorted(table)
    assert {k: lib[k] for k in table} == pkg._OPTION_DEFAULTS
    header = open(os.path.join(root, "include", "exaadmm_hip.h")).read()
    doc = header[header.index("---- options: A/B switches"):header.index("int exaadmm_set_option")]
    assert set(re.findall(r'^ \*   "(\w+)" \(', doc, flags=re.M)) == table
    # the test build's switches are not options: no key of exaadmm_test_hook is accepted by the shipped exaadmm_set_option
    hooks = src[src.index("const OptKey TEST_KEYS[] = {"):]
    hooks = set(re.findall(r'\{"(\w+)",\s*&Options::', hooks[:hooks.index("};")]))
    assert hooks and not (hooks & table)
