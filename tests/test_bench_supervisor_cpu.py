"""CPU: the supervisor of the multi-GPU benchmark legs (exaadmm.jl_amd/bench_supervisor.py) with FAKE measuring children.

What `bench.py --gpus N` must guarantee on its first hardware run: a child that hangs or fails -- alone or with
its peers -- costs one attempt, not the line; every rank's child is ended with it; the next attempt starts fresh children in the next
mode; the line that arrives says which attempts it took.  Here the children are a small script whose behaviour is scripted per
(attempt, rank); world 1 runs in this process, world 2 as two supervisor processes over a TCP store (what the launcher's rendezvous
gives them on a GPU node).  No GPU, no HIP library.
"""
import json
import os
import subprocess
import sys
import time

import pytest

import exaadmm_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE_CHILD = r'''
import json, os, sys, time
plan = json.loads(os.environ["FAKE_PLAN"])          # {"<leg>:<attempt>:<rank>": "ok" | "hang" | "fail" | "noline"}; default "ok"
leg = sys.argv[sys.argv.index("--leg") + 1] if "--leg" in sys.argv else "main"
attempt, rank = os.environ["EXAADMM_BENCH_ATTEMPT"], os.environ["RANK"]
what = plan.get("%s:%s:%s" % (leg, attempt, rank), "ok")
assert os.environ["EXAADMM_BENCH_CHILD"] == "1" and "TORCHELASTIC_RUN_ID" not in os.environ
sys.stderr.write("fake child leg=%s attempt=%s rank=%s -> %s\n" % (leg, attempt, rank, what))
if what == "hang":
    time.sleep(3600)
if what == "fail":
    sys.stderr.write("boom on rank %s\n" % rank)
    sys.exit(3)
if what == "ok":
    time.sleep(0.3)
    if rank == "0":
        print("some library banner")
        print(json.dumps({"metric": "ADMM inner iterations/sec", "value": 100.0 + 10 * int(attempt), "leg": leg,
                          "p2p": os.environ.get("EXAADMM_DIST_P2P", "1"), "graph": "0" if os.environ.get("EXAADMM_NO_GRAPH") else "1", "overlap": os.environ.get("EXAADMM_DIST_OVERLAP", "1"),
                          "python_loop": os.environ.get("EXAADMM_DIST_PYTHON_LOOP", ""), "port": os.environ["MASTER_PORT"]}))
'''


@pytest.fixture(scope="module")
def sup():
    return exaadmm_loader.load_submodule("bench_supervisor")


def _make_cmd(tmp_path):
    script = tmp_path / "fake_child.py"
    script.write_text(FAKE_CHILD)

    def make_cmd(leg):
        return [sys.executable, str(script)] + (["--leg", leg["name"]] if leg["name"] != "main" else [])
    return make_cmd


def _run1(sup, tmp_path, plan, legs=None, budget=4.0):
    env = dict(os.environ, FAKE_PLAN=json.dumps(plan), TORCHELASTIC_RUN_ID="x")
    logs = []
    t0 = time.time()
    line, rc = sup.supervise(_make_cmd(tmp_path), 0, 1, 0, legs or [{"name": "main"}], budget_s=budget, base_env=env, log=logs.append)
    return line, rc, time.time() - t0, logs


def test_first_attempt_succeeds(sup, tmp_path):
    line, rc, dt, _ = _run1(sup, tmp_path, {})
    assert rc == 0 and line["value"] == 100.0 and line["graph"] == "1" and line["p2p"] == "1"
    assert [a["mode"] for a in line["attempts"]] == ["resident+xgmi"] and line["attempts"][0]["ok"]


def test_a_hanging_child_costs_one_attempt_not_the_line(sup, tmp_path):
    line, rc, dt, logs = _run1(sup, tmp_path, {"main:0:0": "hang"}, budget=2.0)
    assert rc == 0 and line["value"] == 110.0
    a = line["attempts"]
    assert [x["mode"] for x in a] == ["resident+xgmi", "graph+overlap"]
    assert a[0]["rc"] != 0 and a[0]["timed_out"] and not a[0]["ok"] and a[1]["ok"]
    assert line["p2p"] == "0" and line["graph"] == "1"        # the second mode's switches reached the fresh child
    assert 2.0 <= dt < 20.0


def test_failures_walk_down_the_modes_and_the_stderr_tail_is_kept(sup, tmp_path):
    line, rc, dt, _ = _run1(sup, tmp_path, {"main:0:0": "fail", "main:1:0": "fail", "main:2:0": "noline", "main:3:0": "fail"})
    assert rc == 0
    a = line["attempts"]
    assert [x["mode"] for x in a] == ["resident+xgmi", "graph+overlap", "overlap", "one-stream", "python-twin"]
    assert a[0]["rc"] == 3 and "boom on rank 0" in a[0]["stderr_tail"]
    assert not a[2]["ok"] and "without a result line" in a[2]["stderr_tail"]
    assert line["python_loop"] == "1" and line["value"] == 140.0


def test_nothing_works_gives_a_line_that_says_so(sup, tmp_path):
    plan = {"main:%d:0" % k: "fail" for k in range(8)}
    line, rc, dt, _ = _run1(sup, tmp_path, plan)
    assert rc == 1 and line["value"] is None and len(line["attempts"]) == len(sup.MODES)
    assert all(not x["ok"] for x in line["attempts"])


def test_an_extra_leg_is_merged_and_its_failure_costs_nothing_else(sup, tmp_path):
    legs = [{"name": "main"}, {"name": "north_star_scaling", "budget_s": 2.0}]
    line, rc, _, _ = _run1(sup, tmp_path, {"main:0:0": "fail"}, legs=legs)
    assert rc == 0 and line["value"] == 110.0
    ns = line["north_star_scaling"]
    assert ns["leg"] == "north_star_scaling" and ns["p2p"] == "0"     # starts in the mode the main leg succeeded in
    assert [x["mode"] for x in ns["attempts"]] == ["graph+overlap"]
    line, rc, _, _ = _run1(sup, tmp_path, {"north_star_scaling:0:0": "hang", "north_star_scaling:1:0": "fail"}, legs=legs)
    assert rc == 0 and line["value"] == 100.0
    assert "error" in line["north_star_scaling"] and len(line["north_star_scaling"]["attempts"]) == 2


WORLD2 = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import exaadmm_loader
from datetime import timedelta
from torch.distributed import TCPStore, PrefixStore
sup = exaadmm_loader.load_submodule("bench_supervisor")
rank = int(sys.argv[1])
store = PrefixStore("t", TCPStore("127.0.0.1", %(port)d, 2, is_master=(rank == 0), timeout=timedelta(seconds=60)))
env = dict(os.environ, FAKE_PLAN=%(plan)r)
def make_cmd(leg):
    return [sys.executable, %(script)r]
t0 = time.time()
line, rc = sup.supervise(make_cmd, rank, 2, rank, [{"name": "main"}], budget_s=%(budget)f, base_env=env, store=store, log=lambda s: sys.stderr.write(s + "\n"))
print("RESULT " + json.dumps({"rank": rank, "rc": rc, "line": line, "seconds": time.time() - t0}))
'''


def _run2(tmp_path, plan, budget=6.0):
    import socket
    script = tmp_path / "fake_child.py"
    script.write_text(FAKE_CHILD)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = WORLD2 % {"root": ROOT, "port": port, "plan": json.dumps(plan), "script": str(script), "budget": budget}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, (o[-1500:], e[-3000:])
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    return outs


def test_world_two__one_rank_fails_and_the_other_is_not_left_hanging(tmp_path):
    """Rank 1's child dies at once in attempt 0 while rank 0's would wait for it for ever (a collective): rank 0's supervisor sees the
    flag, ends its child after the grace period -- long before the budget -- and both move on to the next mode together."""
    outs = _run2(tmp_path, {"main:0:1": "fail", "main:0:0": "hang"}, budget=60.0)
    r0, r1 = outs
    assert r0["rc"] == 0 and r1["rc"] == 0
    a = r0["line"]["attempts"]
    assert [x["mode"] for x in a] == ["resident+xgmi", "graph+overlap"] and a[1]["ok"]
    assert a[0].get("ended_because_a_peer_failed") and a[0]["failed_ranks"] == 1
    assert r0["line"]["value"] == 110.0 and r0["seconds"] < 30.0 and r1["seconds"] < 30.0
    assert r1["line"] == {}            # only rank 0 carries the line


def test_world_two__a_hang_on_one_rank_ends_both_children_at_the_budget(tmp_path):
    outs = _run2(tmp_path, {"main:0:1": "hang"}, budget=3.0)
    a = outs[0]["line"]["attempts"]
    assert [x["ok"] for x in a] == [False, True] and outs[0]["line"]["value"] == 110.0
    assert all(o["rc"] == 0 and o["seconds"] < 40.0 for o in outs)


def test_result_line_is_the_last_json_object_with_a_metric(sup):
    txt = 'RCCL version 2.26\n{"x": 1}\n{"metric": "m", "value": 1}\ntrailing noise\n'
    assert sup.result_line(txt) == {"metric": "m", "value": 1}
    assert sup.result_line("nothing here\n") is None


def test_the_real_launcher_path_on_a_box_without_gpus():
    """`python bench.py --gpus 2` as the driver's launcher runs it (torch.distributed.run -> one supervising bench.py per rank -> the ranks'
    supervisors meet in the launcher's own rendezvous store -> children) on THIS box, which has no GPU: every child dies at its first HIP call
    in every mode, and what must still come back is ONE JSON line -- value null, the six attempts with their modes and the children's error --
    and a non-zero exit code, within seconds (nobody waits for a budget: the failure flag travels through the store)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the children would measure")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--attempt-budget", "120",
                        "--no-north-star"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode != 0
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 2 and "error" in line
    assert [a["mode"] for a in line["attempts"]] == ["resident+xgmi", "graph+overlap", "overlap", "one-stream", "python-twin", "python-gloo"]
    assert all(not a["ok"] and a["failed_ranks"] == 2 and "No HIP GPUs" in a["stderr_tail"] for a in line["attempts"]), line["attempts"]
    assert time.time() - t0 < 300
