"""-m gpu: the resident inner loop (kernels.h k_resident: ONE launch per batch of iterations, every wavefront resident, a bus
updated by the last lane to arrive at it, residual sums and the exact per-iteration stop test made by a control wavefront)
against the two-launch loop (EXAADMM_NO_RESIDENT=1).  Every lane does the arithmetic it does in the two-launch loop and every
sum runs in the same order, so everything must be BITWISE equal: u, v, z, l, multipliers / penalties, per-branch counters, the
four norms, the iteration at which the convergence-tested loop stops, whole solves.  Grids with more wavefronts than the
chip has SIMDs must fall back to the two-launch loop by themselves."""
import os

import numpy as np
import pytest

import exaadmm_loader

pytestmark = pytest.mark.gpu
SWITCHED_OFF = bool(os.environ.get("EXAADMM_NO_SPEC") or os.environ.get("EXAADMM_BUS_KERNEL") == "lane" or os.environ.get("EXAADMM_NO_REC")
                    or os.environ.get("EXAADMM_NO_RESIDENT") or os.environ.get("EXAADMM_RESIDENT") == "0")


@pytest.fixture(scope="module")
def pkg():
    p = exaadmm_loader.load_package()
    p.library()
    return p


class _two_launch:
    """environment switch for the duration of a block (the library reads it when a loop starts)"""
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("EXAADMM_NO_RESIDENT")
        if self.on:
            os.environ["EXAADMM_NO_RESIDENT"] = "1"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("EXAADMM_NO_RESIDENT", None)
        else:
            os.environ["EXAADMM_NO_RESIDENT"] = self.old


def _state(mod):
    s = mod.solution
    a, m, t = mod.branch_counters()
    return [s.u_curr.copy(), s.v_curr.copy(), s.z_curr.copy(), s.l_curr.copy(), np.ascontiguousarray(mod.membuf).copy(), a, m, t]


@pytest.mark.skipif(SWITCHED_OFF, reason="a switch of this run turns the resident loop off")
@pytest.mark.parametrize("case,rho", [("case1354pegase", (1e1, 1e3)), ("case13659pegase", (5e1, 5e3)), ("case_ACTIVSg70k", (3e4, 3e5))])
def test_resident_loop_is_bitwise_the_two_launch_loop(pkg, case, rho):
    """(case_ACTIVSg70k's size, 1 379 tiles of 64 branches + 163 generator tiles: more than the chip has SIMDs -- the resident
    wavefronts own up to TWO tiles each there, k_resident_t<2>, assigned per launch from the solver-trip counters)"""
    if case == "case_ACTIVSg70k" and os.environ.get("EXAADMM_RESIDENT_TILES") == "0":
        pytest.skip("a switch of this run turns the two-tile form off")
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like(case, seed=0)
    runs = []
    for two in (True, False):
        with _two_launch(two):
            env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
            mod = pkg.ModelAcopf(env)
            assert mod.loop_kind() == (1 if two else 2)
            out, it = [], 1
            for K in (1, 2, 13, 60, 33, 2, 5, 40):  # odd and even lengths, odd and even starting iterations; from the loop's 65th
                                                    # iteration on the launches have heavy wavefronts (kernels.h k_resident_t<1, true>)
                norms, _, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
                it += K
                out.append(_state(mod) + [norms.copy()])
            for outer, limit in ((1, 45), (2, 7), (3, 200)):   # convergence-tested loop: stops by the limit and by the test, several batches
                inner, res = mod.run_inner(outer, limit, 1e3)
                out.append(_state(mod) + [res.copy(), np.array([inner])])
            runs.append(out)
            mod.close()
    for a, b in zip(runs[0], runs[1]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


@pytest.mark.skipif(SWITCHED_OFF or bool(os.environ.get("EXAADMM_NO_HEAVY")), reason="a switch of this run turns the resident loop / its heavy wavefronts off")
@pytest.mark.parametrize("case,rho", [("case1354pegase", (1e1, 1e3)), ("case13659pegase", (5e1, 5e3))])
def test_heavy_wavefronts_leave_every_bit_where_the_lanes_leave_it(pkg, case, rho):
    """Launches that continue an inner loop beyond its 64th iteration give the branches with the most solver trips a wavefront of
    their own (64 clones, the Cauchy search's candidates side by side; kernels.h k_resident_t<1, true>) -- against the same
    resident loop without them (EXAADMM_NO_HEAVY=1): fixed-count runs of many lengths, and a convergence-tested loop that needs a
    second launch (1 024 iterations per launch), everything bitwise equal."""
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like(case, seed=0)
    runs = []
    for off in (True, False):
        old = os.environ.get("EXAADMM_NO_HEAVY")
        if off:
            os.environ["EXAADMM_NO_HEAVY"] = "1"
        try:
            env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
            mod = pkg.ModelAcopf(env)
            assert mod.loop_kind() == 2
            out, it = [], 1
            for K in (70, 1, 2, 3, 4, 5, 8, 31, 64):
                norms, _, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
                it += K
                out.append(_state(mod) + [norms.copy()])
            inner, res = mod.run_inner(1000000, 1100, 1e3)   # (never converges by its test: two launches, the second one with heavy wavefronts)
            out.append(_state(mod) + [res.copy(), np.array([inner])])
            assert inner == 1100 and mod.loop_downgrades() == 0
            runs.append(out)
            mod.close()
        finally:
            if old is None:
                os.environ.pop("EXAADMM_NO_HEAVY", None)
            else:
                os.environ["EXAADMM_NO_HEAVY"] = old
    for a, b in zip(runs[0], runs[1]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert runs[0][-2][7].max() >= 3   # (there were branches with three and more trips to take)


@pytest.mark.skipif(SWITCHED_OFF, reason="a switch of this run turns the resident loop off")
def test_resident_whole_solves_reproduce_the_anchors_and_the_two_launch_runs(pkg):
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for case, outer, cumul in (("case9.m", 20, 705), ("case118.m", 20, 1232)):
        infos = []
        for two in (True, False):
            with _two_launch(two):
                env, mod = pkg.solve_acopf(os.path.join(golden, case), outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5, verbose=0)
                i = mod.info
                infos.append((i.status, i.outer, i.cumul, i.objval, i.primres, i.dualres, i.mismatch, mod.loop_kind(),
                              mod.solution.u_curr.copy(), mod.solution.v_curr.copy(), mod.solution.l_curr.copy()))
                mod.close()
        assert infos[0][:3] == ("Solved", outer, cumul) == infos[1][:3]
        assert (infos[0][7], infos[1][7]) == (1, 2)
        assert infos[0][3:7] == infos[1][3:7]
        for k in (8, 9, 10):
            assert np.array_equal(infos[0][k], infos[1][k])
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like("case13659pegase", seed=0)
    res = []
    for two in (True, False):
        with _two_launch(two):
            env, mod = pkg.solve_acopf("case13659pegase", grid=g, rho_pq=5e1, rho_va=5e3, outer_iterlim=20, inner_iterlim=1000, verbose=0)
            i = mod.info
            res.append((i.status, i.outer, i.cumul, i.objval, i.primres, mod.solution.u_curr.tobytes(), mod.solution.v_curr.tobytes()))
            mod.close()
    assert res[0] == res[1] and res[0][:3] == ("Solved", 3, 250)


@pytest.mark.skipif(SWITCHED_OFF or bool(os.environ.get("EXAADMM_NO_HEAVY")), reason="a switch of this run turns the resident loop / its heavy wavefronts off")
@pytest.mark.parametrize("case,rho", [("case1354pegase", (1e1, 1e3)), ("case13659pegase", (5e1, 5e3))])
def test_stragglers_adopted_by_free_heavy_wavefronts_inside_a_launch(pkg, case, rho):
    """Option heavy_adopt = t: a branch lane whose solve took >= t trips hands its branch to a free heavy wavefront from the next
    iteration on (kernels.h ResidentHeavy: adoption) -- the state travels through an agent-scope record, the planes' words of the
    branch stay the lane's (it writes them from the heavy wavefront's hand-over).  Off, at the default and at 3 trips (many more
    adoptions), launches of odd and even lengths: every bit the same."""
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like(case, seed=0)
    runs = []
    for adopt in (0, 4, 3):
        with _env(EXAADMM_HEAVY_ADOPT=adopt):
            env = pkg.AdmmEnv(case, rho[0], rho[1], verbose=0, grid=g)
            mod = pkg.ModelAcopf(env)
            assert mod.loop_kind() == 2
            out, it = [], 1
            for K in (70, 3, 40, 1, 64, 33):
                norms, _, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
                it += K
                out.append(_state(mod) + [norms.copy()])
            assert mod.loop_downgrades() == 0, pkg.library().exaadmm_last_error()
            runs.append(out)
            mod.close()
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)


@pytest.mark.skipif(SWITCHED_OFF, reason="a switch of this run turns the resident loop off")
def test_fixed_count_runs_longer_than_one_launch_may_be(pkg):
    """exaadmm_run_fixed used to issue all its iterations as ONE resident launch.  The lanes of a launch pack the launch-relative
    iteration index into 32-bit words next to a branch number (adoption requests: index << 20) and into the mailboxes' sequence
    numbers ((epoch << 12) | index): from index 2 048 / 4 096 on those overflow -- a free heavy wavefront would start its adopted
    branch at a wrong iteration and add spurious arrivals at its buses, silently.  A launch now covers at most 1 024
    iterations, as exaadmm_run_inner's always did: 2 100 iterations that continue an inner loop (heavy wavefronts and adoption on,
    three launches) are bitwise the two-launch loop's."""
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like("case1354pegase", seed=0)
    runs = []
    for two in (True, False):
        with _two_launch(two):
            env = pkg.AdmmEnv("case1354pegase", 1e1, 1e3, verbose=0, grid=g)
            mod = pkg.ModelAcopf(env)
            assert mod.loop_kind() == (1 if two else 2)
            n1, _, _ = mod.run_fixed(300, first_major_iter=1, beta=1e3)
            n2, _, _ = mod.run_fixed(2100, first_major_iter=301, beta=1e3)
            if not two:
                f = mod.resident_form()
                assert f["tiles"] == 1 and f["launches"] == 3, f        # 1 024 + 1 024 + 52
                assert mod.loop_downgrades() == 0, pkg.library().exaadmm_last_error()
            runs.append(_state(mod) + [n1.copy(), n2.copy()])
            mod.close()
    for x, y in zip(*runs):
        assert np.array_equal(x, y)


@pytest.mark.skipif(SWITCHED_OFF or bool(os.environ.get("EXAADMM_NO_HEAVY")), reason="a switch of this run turns the resident loop / its heavy wavefronts off")
def test_whole_solves_whose_inner_loops_continue_with_heavy_wavefronts(pkg):
    """Option heavy_split = S: the first launch of every inner loop ends after S iterations and the rest of the loop runs with heavy
    wavefronts -- in loops that END BY THEIR STOP TEST.  A heavy wavefront does not wait for the stop decision before its solve
    (kernels.h): one that returns on a stop has not arrived at its buses, and one that waits for a bus must hear of the stop too
    (met: a capped wait and a downgrade on the first whole solve that ran this way).  Same anchors, same bits as the default
    launches, no downgrade, on grids of 1, 3, 32 and 320 tiles."""
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    syn = exaadmm_loader.load_submodule("synthetic")
    cases = [(os.path.join(golden, "case9.m"), None, dict(outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5), ("Solved", 20, 705)),
             (os.path.join(golden, "case118.m"), None, dict(outer_iterlim=25, rho_pq=4e2, rho_va=4e4, outer_eps=2e-5), ("Solved", 20, 1232)),
             ("case1354pegase", syn.sized_like("case1354pegase", seed=0), dict(rho_pq=1e1, rho_va=1e3, outer_iterlim=20, inner_iterlim=1000), None),
             ("case13659pegase", syn.sized_like("case13659pegase", seed=0), dict(rho_pq=5e1, rho_va=5e3, outer_iterlim=20, inner_iterlim=1000), ("Solved", 3, 250))]
    for case, g, kw, anchor in cases:
        res = []
        for split in (0, 8, 21):
            with _env(EXAADMM_HEAVY_SPLIT=split, EXAADMM_HEAVY_SPLIT_PREV=0):
                env, mod = pkg.solve_acopf(case, verbose=0, **(dict(kw, grid=g) if g is not None else kw))
                i = mod.info
                assert mod.loop_kind() == 2 and mod.loop_downgrades() == 0, (case, split, pkg.library().exaadmm_last_error())
                res.append((i.status, i.outer, i.cumul, i.objval, i.primres, i.dualres, mod.solution.u_curr.tobytes(), mod.solution.v_curr.tobytes(),
                            mod.solution.l_curr.tobytes()))
                if split and g is not None:
                    assert mod.resident_form()["heavy"] > 0, (case, split)   # (the last launch of the solve carried heavy wavefronts; the two small
                                                                             #  cases end on inner loops shorter than the split)
                mod.close()
        assert res[0] == res[1] == res[2], case
        if anchor:
            assert res[0][:3] == anchor


def test_grids_with_more_than_twice_as_many_tiles_as_simds_keep_the_two_launch_loop(pkg):
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.synthetic_grid(20000, 160000, 110000, seed=3)      # 2 500 tiles of 64 branches: not even two per wavefront fit
    env = pkg.AdmmEnv("big", 5e1, 5e3, verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    assert mod.loop_kind() in (0, 1)
    mod.close()
    if not SWITCHED_OFF and os.environ.get("EXAADMM_RESIDENT_TILES") != "0":
        g = syn.sized_like("case_ACTIVSg70k", seed=0)           # 1 379 + 163 tiles: two per wavefront
        env = pkg.AdmmEnv("case_ACTIVSg70k", 3e4, 3e5, verbose=0, grid=g)
        mod = pkg.ModelAcopf(env)
        assert mod.loop_kind() == 2
        mod.set_option("resident_tiles", 0)   # (A/B switch of the two-tile form)
        assert mod.loop_kind() == 1
        mod.close()


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# ---- fault injection: compiled into the TEST build of the library only (csrc/Makefile libexaadmm_hip_hooks.so, -DEXA_TEST_HOOKS:
#      exaadmm_test_hook); the scenarios run in a worker process that loads that build
_HOOKS_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package()
L = pkg.library()
assert hasattr(L, "exaadmm_test_hook") and "-DEXA_TEST_HOOKS" in L.exaadmm_version().decode()
syn = exaadmm_loader.load_submodule("synthetic")

def state(mod):
    s = mod.solution
    a, m, t = mod.branch_counters()
    return [s.u_curr.copy(), s.v_curr.copy(), s.z_curr.copy(), s.l_curr.copy(), np.ascontiguousarray(mod.membuf).copy(), a, m, t]

def same(x, y):
    assert len(x) == len(y)
    for p, q in zip(x, y):
        assert np.array_equal(p, q)

def model(grid, **opts):
    env = pkg.AdmmEnv("case1354pegase", 1e1, 1e3, verbose=0, grid=grid)
    mod = pkg.ModelAcopf(env)
    for k, v in opts.items():
        mod.set_option(k, v)
    return mod

report = {}
g = syn.sized_like("case1354pegase", seed=0)
# (1) a resident launch that reports a failed wait: the first batch of a loop / a later batch of a convergence-tested loop (a resident
#     launch covers up to 1 024 iterations: the tolerance of a (hypothetical) millionth outer iteration keeps the loop going into a second
#     launch, which is where fail_at = 2 strikes).  The call restores the iterate it started from, repeats its loop with two launches per
#     iteration and returns what it would have returned; the handle keeps that loop.
for fail_at in (1, 2):
    runs = []
    for inject in (False, True):
        mod = model(g, **({"trust_resident": 1} if inject else {"resident": 0}))
        if inject:
            mod.test_hook("fail_resident_launch", fail_at)
        assert mod.loop_kind() == (2 if inject else 1) and mod.loop_downgrades() == 0
        out = []
        inner, res = mod.run_inner(1000000 if fail_at == 2 else 1, 1100 if fail_at == 2 else 150, 1e3)
        assert fail_at == 1 or inner > 1024
        out += state(mod) + [res.copy(), np.array([inner])]
        norms, _, _ = mod.run_fixed(21, first_major_iter=inner + 1, beta=1e3)
        out += state(mod) + [norms.copy()]
        if inject:
            assert mod.loop_downgrades() == 1 and mod.loop_kind() == 1
        runs.append(out)
        mod.close()
    same(runs[0], runs[1])
report["failed_batches"] = "ok"
# (2) ... in a fixed-count run and in a whole solve
g1 = syn.sized_like("case1354pegase", seed=1)
res = []
for inject in (False, True):
    mod = model(g1, **({"trust_resident": 1} if inject else {"resident": 0}))
    if inject:
        mod.test_hook("fail_resident_launch", 1)
    norms, _, _ = mod.run_fixed(37, first_major_iter=1, beta=1e3)
    st = state(mod) + [norms.copy()]
    assert mod.loop_downgrades() == (1 if inject else 0)
    mod.close()
    env2 = pkg.AdmmEnv("case1354pegase", 1e1, 1e3, verbose=0, grid=g1)
    mod2 = pkg.ModelAcopf(env2)
    mod2.set_option("trust_resident" if inject else "resident", 1 if inject else 0)
    if inject:
        mod2.test_hook("fail_resident_launch", 1)
    env2.params.outer_iterlim, env2.params.inner_iterlim = 3, 200
    pkg.admm_two_level(env2, mod2)
    i = mod2.info
    st += [np.array([i.outer, i.cumul]), np.array([i.objval, i.primres, i.mismatch]), mod2.solution.u_curr.copy(), mod2.solution.v_curr.copy()]
    assert mod2.loop_downgrades() == (1 if inject else 0)
    mod2.close()
    res.append(st)
same(res[0], res[1])
report["failed_fixed_and_solve"] = "ok"
# (3) the first-use self check (exaadmm_hip.h exaadmm_resident_selfcheck_stats): one ulp flipped in ONE word of the resident leg's u must be
#     found -- the handle is downgraded, the error text names the vector, and the call returns the two-launch loop's bits
for entry in ("run_fixed", "run_inner"):
    runs = []
    for inject in (False, True):
        mod = model(g, **({} if inject else {"resident": 0}))
        if inject:
            mod.test_hook("flip_resident_ulp", 1)
            assert mod.loop_kind() == 2
        if entry == "run_fixed":
            norms, _, _ = mod.run_fixed(30, first_major_iter=1, beta=1e3)
            out = state(mod) + [norms.copy()]
        else:
            inner, r4 = mod.run_inner(1, 150, 1e3)
            out = state(mod) + [r4.copy(), np.array([inner])]
        if inject:
            assert mod.resident_selfchecks() == (1, 1), mod.resident_selfchecks()
            assert mod.loop_downgrades() == 1 and mod.loop_kind() == 1
            msg = L.exaadmm_last_error().decode()
            assert "self check" in msg and "u: 1 word" in msg, msg
            report["message_" + entry] = msg
        norms, _, _ = mod.run_fixed(9, first_major_iter=200, beta=1e3)   # ... and goes on with the two-launch loop
        out += state(mod) + [norms.copy()]
        runs.append(out)
        mod.close()
    same(runs[0], runs[1])
report["flipped_ulp_found"] = "ok"
print("RESULT " + json.dumps(report))
'''


@pytest.mark.skipif(SWITCHED_OFF, reason="a switch of this run turns the resident loop off")
def test_injected_failures_and_an_injected_ulp_are_caught__test_build_of_the_library():
    """The resident loop's only failure mode -- a wavefront of the launch is not on the chip (shared GPU), a wait hits its cap -- must
    not surface, and a resident kernel that does not reproduce the two-launch loop bit for bit on this device must be found by the
    first-use self check before anything it produced is returned.  Both are INJECTED (a launch that reports a failed wait; one ulp
    added to one word of the self check's resident leg) through exaadmm_test_hook, which exists only in the test build of the library:
    the scenarios run in a worker process that loads libexaadmm_hip_hooks.so."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    lib = os.path.join(ROOT, "exaadmm.jl_amd", "libexaadmm_hip_hooks.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "exaadmm.jl_amd", "csrc"), "../libexaadmm_hip_hooks.so"])
    env = dict(os.environ, EXAADMM_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-c", _HOOKS_WORKER % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(rep)
    assert rep["failed_batches"] == rep["failed_fixed_and_solve"] == rep["flipped_ulp_found"] == "ok"


@pytest.mark.skipif(SWITCHED_OFF, reason="a switch of this run turns the resident loop off")
def test_first_use_self_check_runs_once_per_kernel_form_and_leaves_the_iterate_alone(pkg):
    """Default library: the first call that used a resident kernel form is followed by ONE self check of that form (8 iterations from the
    call's result with the resident and with the two-launch loop, compared on the device, the iterate restored): no mismatch on this
    hardware, the results of the checked calls bitwise those of a handle that trusts the loop (option "trust_resident"), and no
    further checks afterwards."""
    syn = exaadmm_loader.load_submodule("synthetic")
    # (a size no other test of this process uses: a form that has passed for a launch geometry on this device is not checked again in the
    #  same process -- which the third handle below shows)
    g = syn.synthetic_grid(4100, 20400, 13600, seed=0)
    runs, checks = [], []
    for trust in (0, 1, 0):
        env = pkg.AdmmEnv("case13659pegase", 5e1, 5e3, verbose=0, grid=g)
        mod = pkg.ModelAcopf(env)
        mod.set_option("trust_resident", trust)
        assert mod.loop_kind() == 2
        out, seen = [], []
        n1, _, _ = mod.run_fixed(40, first_major_iter=1, beta=1e3)      # the plain kernel form
        out += _state(mod) + [n1.copy()]
        seen.append(mod.resident_selfchecks())
        f1 = mod.resident_form()
        n2, _, _ = mod.run_fixed(30, first_major_iter=80, beta=1e3)     # a launch that continues beyond the 64th iteration: heavy wavefronts
        out += _state(mod) + [n2.copy()]
        seen.append(mod.resident_selfchecks())
        f2 = mod.resident_form()
        n3, _, _ = mod.run_fixed(30, first_major_iter=110, beta=1e3)
        inner, res = mod.run_inner(1, 60, 1e3)
        out += _state(mod) + [n3.copy(), res.copy(), np.array([inner])]
        seen.append(mod.resident_selfchecks())
        assert mod.loop_downgrades() == 0 and mod.loop_kind() == 2
        assert f1["tiles"] == 1 and f1["heavy"] == 0 and f1["launches"] == 1 and f1["wavefronts"] >= 385
        if not os.environ.get("EXAADMM_NO_HEAVY"):
            assert f2["heavy"] > 0 and f2["wavefronts"] == f1["wavefronts"] + f2["heavy"]
        runs.append(out)
        checks.append(seen)
        mod.close()
    for other in runs[1:]:
        for x, y in zip(runs[0], other):
            assert np.array_equal(x, y)
    assert checks[1] == [(0, 0)] * 3 and checks[2] == [(0, 0)] * 3
    assert checks[0] == ([(1, 0), (2, 0), (2, 0)] if not os.environ.get("EXAADMM_NO_HEAVY") else [(1, 0), (1, 0), (1, 0)]), checks[0]


_CONTENDER = r'''
import hashlib, json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import exaadmm_loader
pkg = exaadmm_loader.load_package(); pkg.library()
syn = exaadmm_loader.load_submodule("synthetic")
g = syn.sized_like("case13659pegase", seed=0)
env = pkg.AdmmEnv("case13659pegase", 5e1, 5e3, verbose=0, grid=g)
mod = pkg.ModelAcopf(env)
kind0 = mod.loop_kind()
open(%(flag)r + str(os.getpid()), "w").write("ready")
while len([f for f in os.listdir(os.path.dirname(%(flag)r)) if f.startswith(os.path.basename(%(flag)r))]) < %(n)d:
    time.sleep(0.01)                       # all contenders start their loops together
t0 = time.time()
h = hashlib.sha256()
it = 1
for K in (300, 300, 300):
    norms, _, _ = mod.run_fixed(K, first_major_iter=it, beta=1e3)
    it += K
    for v in (mod.solution.u_curr, mod.solution.v_curr, mod.solution.z_curr, mod.solution.l_curr, norms):
        h.update(np.ascontiguousarray(v).tobytes())
print("RESULT " + json.dumps({"digest": h.hexdigest(), "kind_before": kind0, "kind_after": mod.loop_kind(),
                              "downgrades": mod.loop_downgrades(), "seconds": time.time() - t0}))
'''


@pytest.mark.skipif(SWITCHED_OFF, reason="a switch of this run turns the resident loop off")
def test_three_processes_share_the_gpu_and_all_finish_with_the_same_bits(tmp_path):
    """Three processes run the 13 659-bus grid on ONE GPU at the same time: 3 x 385 wavefronts do not fit 1 024 SIMDs, so not
    every resident launch can have all its wavefronts on the chip.  Whatever the hardware scheduler does -- it may serialise the
    launches (everybody stays resident) or interleave them (waits hit their cap: downgrade) -- every process must finish, with
    status OK and the bits of an undisturbed run."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    flag = str(tmp_path / "ready_")
    code = _CONTENDER % {"root": ROOT, "flag": flag, "n": 3}
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(3)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, (o[-1500:], e[-3000:])
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    alone = subprocess.run([sys.executable, "-c", _CONTENDER % {"root": ROOT, "flag": str(tmp_path / "alone_"), "n": 1}],
                           capture_output=True, text=True, timeout=900)
    assert alone.returncode == 0, alone.stderr[-3000:]
    ref = json.loads([l for l in alone.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert ref["kind_before"] == 2 and ref["downgrades"] == 0
    print("contended runs:", outs, "alone:", ref)
    for o in outs:
        assert o["digest"] == ref["digest"], (o, ref)
        assert o["kind_before"] == 2 and o["kind_after"] == (1 if o["downgrades"] else 2)


def _hog_binary():
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "tools", "gpu_hog")
    src = exe + ".hip"
    if not os.path.exists(exe) or os.path.getmtime(src) > os.path.getmtime(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", exe, src])
    return exe


@pytest.mark.skipif(SWITCHED_OFF, reason="a switch of this run turns the resident loop off")
def test_a_gpu_mostly_taken_by_another_process_downgrades_the_loop_not_the_call(tmp_path):
    """Another process holds 200 of the 256 compute units (one block per CU with all of its LDS) for a few seconds while the
    13 659-bus grid (385 wavefronts) starts its resident loop.  Two things can happen, and both have been observed on this pool:
    the dispatcher holds the (cooperative) launch back until the other process lets go -- the call just takes longer -- or it
    places the 4 x 56 wavefronts that fit, the others never start, waits hit their cap, and the library falls back to the
    two-launch loop.  Either way the call must come back OK with the bits of an undisturbed run."""
    import json
    import subprocess
    import sys
    import time
    from conftest import ROOT
    alone = subprocess.run([sys.executable, "-c", _CONTENDER % {"root": ROOT, "flag": str(tmp_path / "alone_"), "n": 1}],
                           capture_output=True, text=True, timeout=900)
    assert alone.returncode == 0, alone.stderr[-3000:]
    ref = json.loads([l for l in alone.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    # the contender waits (flag files) until the hog is running: the hog's flag is written here once it says so
    flag = str(tmp_path / "go_")
    cont = subprocess.Popen([sys.executable, "-c", _CONTENDER % {"root": ROOT, "flag": flag, "n": 2}], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True)
    while not [f for f in os.listdir(tmp_path) if f.startswith("go_")]:      # contender has built its model
        assert cont.poll() is None, cont.communicate()
        time.sleep(0.05)
    hog = subprocess.Popen([_hog_binary(), "200", "4.0"], stdout=subprocess.PIPE, text=True)
    assert "HOG RUNNING" in hog.stdout.readline()
    time.sleep(0.3)
    open(flag + "hog", "w").write("ready")
    o, e = cont.communicate(timeout=900)
    hog_out = hog.communicate(timeout=60)[0]
    assert cont.returncode == 0, (o[-1500:], e[-3000:])
    assert "HOG DONE" in hog_out and hog.returncode == 0, hog_out
    got = json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:])
    print("beside the hog:", got, "alone:", ref)
    assert got["digest"] == ref["digest"]
    assert (got["downgrades"], got["kind_after"]) in ((1, 1), (0, 2)), got


def test_options_replace_the_environment_switches(pkg):
    """exaadmm_set_option / exaadmm_get_option (include/exaadmm_hip.h): the library reads no environment variable; unknown keys and values out
    of range are errors; the loop a handle uses follows its options; the Python mirror's translation of EXAADMM_* variables happens when a model
    is created."""
    syn = exaadmm_loader.load_submodule("synthetic")
    g = syn.sized_like("case1354pegase", seed=0)
    env = pkg.AdmmEnv("case1354pegase", 1e1, 1e3, verbose=0, grid=g)
    mod = pkg.ModelAcopf(env)
    want = pkg.options_from_environment()
    for key, value in want.items():
        assert mod.get_option(key) == value, key
    if not SWITCHED_OFF:
        assert mod.loop_kind() == 2
        mod.set_option("resident", 0)
        assert mod.loop_kind() == 1 and mod.get_option("resident") == 0
        mod.set_option("spec", 0)
        assert mod.loop_kind() == 0
        mod.set_option("spec", 1)
        mod.set_option("resident", 1)
        assert mod.loop_kind() == 2
        mod.set_option("rec", 0)               # (no pushed records: the resident loop has nothing to read its terms from)
        assert mod.loop_kind() == 1
        mod.set_option("rec", 1)
    with pytest.raises(pkg.ExaAdmmError, match="unknown option"):
        mod.set_option("no_such_switch", 1)
    with pytest.raises(pkg.ExaAdmmError, match="out of range"):
        mod.set_option("heavy_max", 100000)
    with pytest.raises(pkg.ExaAdmmError):
        mod.get_option("no_such_switch")
    old = os.environ.get("EXAADMM_NO_HEAVY")
    os.environ["EXAADMM_NO_HEAVY"] = "1"
    try:
        m2 = pkg.ModelAcopf(env)
        assert m2.get_option("heavy") == 0 and mod.get_option("heavy") == want["heavy"]
        m2.close()
    finally:
        if old is None:
            os.environ.pop("EXAADMM_NO_HEAVY", None)
        else:
            os.environ["EXAADMM_NO_HEAVY"] = old
    mod.close()
