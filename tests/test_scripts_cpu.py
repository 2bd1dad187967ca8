"""CPU: the scripts that turn rocprofv3 output into what profiles/ holds classify the launches of a bench.py run the way the run made them
(VERDICT r05 "evidence bugs": round 5's kernel_trace_split.py counted the resident launches from the end and called the 8-iteration self check
of the heavy kernel form "warm_up")."""
import csv
import json
import os
import subprocess
import sys

from conftest import ROOT

SPLIT = os.path.join(ROOT, "scripts", "kernel_trace_split.py")


def _trace(path, launches):
    """launches: (kernel name, microseconds) in launch order"""
    t = 1_000_000
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader()
        for name, us in launches:
            w.writerow({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + int(us * 1000)})
            t += int(us * 1000) + 5000


PLAIN = "void exa::k_resident_t<1, false, false>(exa::Dev const*, ...)"
HEAVY = "void exa::k_resident_t<1, true, false>(exa::Dev const*, ...)"


def _run(trace, *args):
    r = subprocess.run([sys.executable, SPLIT, trace] + [str(a) for a in args], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout)["k_resident"]


def test_trace_of_a_run_that_times_a_state_on_the_solve_trajectory(tmp_path):
    """bench.py --state solve --repeats 1 --no-deep-loop-leg: outer iteration 1 = two launches (split at 48) + the two first-use self checks,
    60 untimed iterations of outer iteration 2, then warm-up | timed region | event-bracketed pass"""
    f = str(tmp_path / "t.csv")
    _trace(f, [(PLAIN, 6000), (HEAVY, 4000), (PLAIN, 700), (HEAVY, 600), (PLAIN, 8000), (HEAVY, 400), (HEAVY, 1600), (HEAVY, 1620)])
    k = _run(f, 20, 5, 300, "solve")
    assert k["state_preparation"]["launches"] == 5
    assert k["warm_up"]["iterations"] == 5 and abs(k["warm_up"]["launch_us"] - 400) < 1
    assert k["timed_region"]["iterations"] == 20 and abs(k["timed_region"]["us_per_iteration"] - 80.0) < 0.1
    assert abs(k["event_bracketed_pass"]["us_per_iteration"] - 81.0) < 0.1


def test_trace_of_a_run_that_times_the_deep_loop_state(tmp_path):
    """bench.py --state prep: preparation (plain form) | its self check | warm-up (heavy form) | ITS self check | R regions | event pass --
    the self check of the heavy form must not be taken for the warm-up"""
    f = str(tmp_path / "t.csv")
    _trace(f, [(PLAIN, 15000), (PLAIN, 390), (HEAVY, 900), (HEAVY, 380), (HEAVY, 9300), (HEAVY, 9100), (HEAVY, 9200), (HEAVY, 9250)])
    k = _run(f, 200, 20, 300, "prep")
    assert abs(k["preparation"]["launch_us"] - 15000) < 1 and k["self_check_after_preparation"]["iterations"] == 8
    assert abs(k["warm_up"]["launch_us"] - 900) < 1 and k["warm_up"]["iterations"] == 20
    assert abs(k["self_check_after_warm_up"]["launch_us"] - 380) < 1
    assert k["timed_region"]["repeats"] == 3 and abs(k["timed_region"]["us_per_iteration"] - 46.0) < 0.1     # the median repeat
    assert abs(k["event_bracketed_pass"]["launch_us"] - 9250) < 1


def test_bench_traffic_source_names_where_traffic_came_from(tmp_path, monkeypatch):
    """BENCH_r05.json said `"traffic": 21710344` next to `"traffic_source": "none: ... has no entry for this workload"`: the assignment had moved into
    the fallback of the instruction calibration (VERDICT / ADVICE r05).  pmc_entry keeps the two apart."""
    sys.path.insert(0, ROOT)
    import bench as B
    f = tmp_path / "pmc.json"
    json.dump({"library_version": "lib-1", "commit": "abc", "w1": {"k_resident": {"traffic_bytes": 123, "valu_busy": 0.2, "valu_insts_per_wave_trip": 3000.0}}}, open(f, "w"))
    monkeypatch.setattr(B, "PMC_FILES", (str(f),))
    monkeypatch.setattr(B, "ROOT", "/")
    kb, src, pmc = B.pmc_entry("w1", "k_resident", "lib-1")
    assert kb["traffic_bytes"] == 123 and src.startswith(str(f)) and "lib-1" in src and "abc" in src
    kb, src, _ = B.pmc_entry("w1", "k_resident", "lib-2")          # counters of other code: refused, and the line says so
    assert kb == {} and src.startswith("refused")
    kb, src, pmc = B.pmc_entry("w2", "k_resident", "lib-1")        # no entry: no traffic, and no claim of a source
    assert kb == {} and src.startswith("none") and pmc["library_version"] == "lib-1"
    kb, src, _ = B.pmc_entry("w1", "k_branch", "lib-1")
    assert kb == {} and src.startswith("none")


def test_design_and_readme_quote_what_the_committed_bench_lines_say(tmp_path):
    """scripts/design_numbers.py regenerates the numbers block of DESIGN.md and the figures of README.md's benchmark paragraph from
    profiles/r06_bench*.json: run on a copy of the tree's documents it must leave them as they are committed (nobody edits those numbers by hand,
    and a refreshed batch of profiles without refreshed documents does not pass)."""
    import glob
    import shutil
    root = tmp_path / "tree"
    (root / "scripts").mkdir(parents=True)
    (root / "profiles").mkdir()
    shutil.copy(os.path.join(ROOT, "scripts", "design_numbers.py"), root / "scripts")
    for f in glob.glob(os.path.join(ROOT, "profiles", "r06_bench*.json")) + [os.path.join(ROOT, "profiles", "r06_pmc_summary.json")]:
        shutil.copy(f, root / "profiles")
    for doc in ("DESIGN.md", "README.md"):
        shutil.copy(os.path.join(ROOT, doc), root)
    r = subprocess.run([sys.executable, str(root / "scripts" / "design_numbers.py"), "r06"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    for doc in ("DESIGN.md", "README.md"):
        assert open(root / doc).read() == open(os.path.join(ROOT, doc)).read(), doc + " does not quote profiles/r06_bench*.json (python scripts/design_numbers.py r06)"


def test_the_committed_profiles_were_taken_from_the_committed_sources():
    """The source digest the Makefile embeds in exaadmm_version() (sha256 over the library's sources and headers) is the one the committed PMC summary
    and the committed bench lines name; after a source edit -- a comment is enough -- the test says so by skipping until a new batch of profiles is
    committed (bench.py itself refuses a PMC summary taken from other code).  Bench lines that name another digest than the PMC summary's always fail."""
    import hashlib
    csrc = os.path.join(ROOT, "exaadmm.jl_amd", "csrc")
    files = ["exaadmm_hip.hip", "matpower.cpp", "kernels.h", "branch_solver.h", "genramp_solver.h", "tron_device.h", "dist_native.h", os.path.join("..", "..", "include", "exaadmm_hip.h")]
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "SRC      = exaadmm_hip.hip matpower.cpp" in mk and "HDR      = kernels.h branch_solver.h genramp_solver.h tron_device.h dist_native.h ../../include/exaadmm_hip.h" in mk, "the Makefile's source list changed: update this test"
    digest = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in files)).hexdigest()[:12]
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_summary.json")))
    if not pmc["library_version"].endswith("src:" + digest):
        import pytest   # (work in progress on the sources is not an error of the tree; a committed state should not show this skip)
        pytest.skip("profiles/r06_* were taken from other sources (%s) than the tree's (src:%s): run scripts/final_batch.sh on the GPU box" % (pmc["library_version"][-16:], digest))
    for name in ("r06_bench.json", "r06_bench_driver_command_steps20.json", "r06_bench_case_ACTIVSg70k.json", "r06_bench_synthetic1M.json"):
        line = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert "src:" + digest in line["roofline"]["traffic_source"], (name, line["roofline"]["traffic_source"])
