"""Supervision of the multi-GPU benchmark legs (bench.py --gpus N): the measuring processes are CHILDREN.

`python -m torch.distributed.run ... bench.py --gpus N` starts one bench.py per GPU.  Each of those processes stays a
SUPERVISOR: it never touches the GPU (no HIP call, no torch.cuda call -- so starting and killing children is safe on this pool),
starts the measuring process of its rank as a child in a session of its own, and agrees with the other ranks' supervisors
through the launcher's rendezvous store (TCP, CPU only) on what happened:

  * a child that exits non-zero, or produces no result within the attempt's budget, fails the ATTEMPT for every rank: the flag
    travels through the store, every supervisor ends its own child (process group of the child it started, SIGKILL) -- a rank
    that hangs inside a collective cannot hold the others until the job's own time limit;
  * the next attempt starts FRESH children (new processes, new rendezvous port chosen by rank 0's supervisor, new RCCL
    communicator) in the next, more conservative mode of `MODES`;
  * the first attempt that completes on every rank delivers the line; rank 0 prints it -- ONE JSON line, last on stdout -- with
    `attempts: [{mode, rc, seconds, stderr_tail}]` added, so that a run that needed a fallback says which one and why.

The modes (exaadmm_hip.h: exaadmm_set_option keys, passed to the child as the environment variables the Python mirror translates into options of its handle):
    resident+xgmi   the default path: the resident loop on every rank, what crosses GPUs goes by direct stores into the peer's mailbox
                    (a rank that cannot map a peer's memory makes all ranks take the next mode's loop by itself, inside the child)
    graph+overlap   two launches per iteration; RCCL exchange beside the interior solves, batches incl. exchanges replayed from a hipGraph
    overlap         batches enqueued directly (graph=0)
    one-stream      ... and everything on one stream (dist_overlap=0)
    python-twin     the step-wise twin: Python issues every step, torch.distributed (RCCL) collectives
    python-gloo     the same twin over gloo, staged through the host (no RCCL at all)

Nothing here imports the HIP library or CUDA/HIP parts of torch; tests/test_bench_supervisor_cpu.py drives it with fake children
under two `gloo`-free supervisor processes.  Reference: the upstream solver has no multi-process path to mirror
(/root/reference/src/ExaAdmm.jl:9, src/utils/environment.jl:102) -- this is bench infrastructure of the new multi-GPU loop.
"""
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

MODES = [
    ("resident+xgmi", {}),
    ("graph+overlap", {"EXAADMM_DIST_P2P": "0"}),
    ("overlap", {"EXAADMM_DIST_P2P": "0", "EXAADMM_NO_GRAPH": "1"}),
    ("one-stream", {"EXAADMM_DIST_P2P": "0", "EXAADMM_NO_GRAPH": "1", "EXAADMM_DIST_OVERLAP": "0"}),
    ("python-twin", {"EXAADMM_DIST_PYTHON_LOOP": "1"}),
    ("python-gloo", {"EXAADMM_DIST_PYTHON_LOOP": "1", "EXAADMM_BENCH_BACKEND": "gloo"}),
]


class LocalStore:
    """one supervisor (world 1): the same interface as the rendezvous store"""

    def __init__(self):
        self.kv = {}

    def set(self, k, v):
        self.kv[k] = v if isinstance(v, bytes) else str(v).encode()

    def get(self, k):
        return self.kv[k]

    def add(self, k, n):
        self.kv[k] = str(int(self.kv.get(k, b"0")) + int(n)).encode()
        return int(self.kv[k])


def open_store(rank, world, timeout_s=600):
    """The launcher's own rendezvous (env://: MASTER_ADDR / MASTER_PORT, agent store or rank 0 as host), TCP only."""
    if world <= 1:
        return LocalStore()
    from datetime import timedelta
    from torch.distributed import rendezvous, PrefixStore
    store, _, _ = next(rendezvous("env://", rank=rank, world_size=world, timeout=timedelta(seconds=timeout_s)))
    return PrefixStore("exaadmm_bench_supervisor", store)


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


class _Tail(threading.Thread):
    """drains a child's pipe: keeps every line (stdout) or the tail (stderr, also forwarded to our stderr)"""

    def __init__(self, pipe, keep_all, forward_prefix=None, max_bytes=6000):
        super().__init__(daemon=True)
        self.pipe, self.keep_all, self.prefix, self.max_bytes = pipe, keep_all, forward_prefix, max_bytes
        self.lines = []
        self.size = 0

    def run(self):
        try:
            for raw in iter(self.pipe.readline, b""):
                line = raw.decode("utf-8", "replace")
                self.lines.append(line)
                self.size += len(line)
                if not self.keep_all:
                    while self.size > self.max_bytes and len(self.lines) > 1:
                        self.size -= len(self.lines.pop(0))
                if self.prefix is not None:
                    sys.stderr.write(self.prefix + line)
        except Exception:
            pass

    def text(self):
        return "".join(self.lines)


def _kill_group(proc):
    """end the child we started and everything it started (its own session => its own process group)"""
    try:
        os.killpg(proc.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass
    try:
        proc.wait(timeout=30)
    except Exception:
        pass


def result_line(text):
    """the last line of a child's stdout that is a JSON object with a `metric` (what bench's child prints on rank 0)"""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                obj = json.loads(line)
            except ValueError:
                continue
            if isinstance(obj, dict) and "metric" in obj:
                return obj
    return None


def run_attempt(store, key, rank, world, child_cmd, child_env, budget_s, grace_s=3.0, poll_s=0.2, log=None):
    """One attempt on every rank: start this rank's child, watch it and the attempt's shared failure flag.  Returns
    {ok, rc, seconds, stdout, stderr_tail, failed_ranks, timed_out}; `ok` is the same on every rank."""
    log = log or (lambda s: None)
    if rank == 0:
        store.set(key + "/port", str(free_port()))
    port = int(store.get(key + "/port").decode())
    env = dict(child_env)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    t0 = time.time()
    proc = subprocess.Popen(child_cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL,
                            start_new_session=True)
    out = _Tail(proc.stdout, True)
    err = _Tail(proc.stderr, False, forward_prefix="[rank %d %s] " % (rank, key))
    out.start()
    err.start()
    rc, timed_out, killed_for_peer = None, False, False
    seen_fail_at = None
    while True:
        rc = proc.poll()
        if rc is not None:
            break
        now = time.time()
        if now - t0 > budget_s:
            timed_out = True
            store.add(key + "/fail", 1)
            log("rank %d: %s produced no result in %.0f s: ending the child" % (rank, key, budget_s))
            _kill_group(proc)
            rc = proc.poll()
            break
        if store.add(key + "/fail", 0) > 0:   # another rank's child failed or timed out: ours cannot finish either
            seen_fail_at = seen_fail_at or now
            if now - seen_fail_at > grace_s:
                killed_for_peer = True
                _kill_group(proc)
                rc = proc.poll()
                break
        time.sleep(poll_s)
    seconds = time.time() - t0
    out.join(timeout=5)
    err.join(timeout=5)
    if rc != 0 and not killed_for_peer and not timed_out:
        store.add(key + "/fail", 1)
    store.add(key + "/done", 1)
    # everybody's child has ended (or been ended) before anybody decides: the flag is final then
    t_wait = time.time()
    while store.add(key + "/done", 0) < world:
        if time.time() - t_wait > budget_s + 120:
            store.add(key + "/fail", 1)
            break
        time.sleep(poll_s)
    failed = store.add(key + "/fail", 0)
    return {"ok": failed == 0, "rc": rc if rc is not None else -9, "seconds": seconds, "stdout": out.text(), "stderr_tail": err.text()[-1500:],
            "failed_ranks": failed, "timed_out": timed_out, "ended_for_peer": killed_for_peer}


def child_environment(base_env, mode_env, local_rank):
    env = {k: v for k, v in base_env.items() if not k.startswith("TORCHELASTIC_") and not k.startswith("EXAADMM_OPT_DIST_")}
    env.pop("EXAADMM_NO_GRAPH", None)
    env.pop("EXAADMM_DIST_OVERLAP", None)
    env.pop("EXAADMM_DIST_PYTHON_LOOP", None)
    env.pop("EXAADMM_BENCH_BACKEND", None)
    env.pop("EXAADMM_DIST_P2P", None)
    env.update(mode_env)
    env["EXAADMM_BENCH_CHILD"] = "1"
    env["LOCAL_RANK"] = str(local_rank)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    return env


def supervise(make_cmd, rank, world, local_rank, legs, budget_s=240.0, modes=None, base_env=None, log=None, store=None):
    """Runs the legs one after the other.  make_cmd(leg) -> argv of a child.  legs: [{"name", "required", "budget_s"?}]; the FIRST
    leg is the benchmark line itself (required), later ones are extras merged into it under their name -- an extra that fails in
    every mode it is tried in is reported as {"error": ..., "attempts": [...]} and costs nothing else.
    Returns (line or None, exit code); rank 0 gets the line."""
    modes = modes if modes is not None else MODES
    base_env = dict(os.environ if base_env is None else base_env)
    log = log or (lambda s: sys.stderr.write("bench supervisor: %s\n" % s))
    store = store if store is not None else open_store(rank, world)
    line, first_mode = None, 0
    inject = base_env.get("EXAADMM_BENCH_INJECT", "")   # tests: "hang:<attempt>" / "fail:<attempt>" applied by the child itself
    for li, leg in enumerate(legs):
        attempts, got = [], None
        # an extra leg starts in the mode the main leg succeeded in, and gets at most two modes
        mode_ids = list(range(first_mode, len(modes))) if li == 0 else list(range(first_mode, len(modes)))[:2]
        for ai, mi in enumerate(mode_ids):
            name, mode_env = modes[mi]
            key = "leg%d/attempt%d" % (li, ai)
            env = child_environment(base_env, mode_env, local_rank)
            env["EXAADMM_BENCH_ATTEMPT"] = str(ai)
            if inject:
                env["EXAADMM_BENCH_INJECT"] = inject
            r = run_attempt(store, key, rank, world, make_cmd(leg), env, float(leg.get("budget_s", budget_s)), log=log)
            rec = {"mode": name, "rc": r["rc"], "seconds": round(r["seconds"], 1), "ok": r["ok"], "failed_ranks": r["failed_ranks"]}
            if r["timed_out"]:
                rec["timed_out"] = True
            if r["ended_for_peer"]:
                rec["ended_because_a_peer_failed"] = True
            if not r["ok"]:
                rec["stderr_tail"] = r["stderr_tail"]
            attempts.append(rec)
            if r["ok"]:
                got = result_line(r["stdout"]) if rank == 0 else {}
                if rank == 0 and got is None:
                    # every child exited 0 but rank 0 printed no line: a failed attempt for everybody (agreed through the store)
                    store.add(key + "/noline", 1)
                # (one more agreement round: rank 0 says whether it has a line)
                store.add(key + "/checked", 1)
                t0 = time.time()
                while store.add(key + "/checked", 0) < world and time.time() - t0 < 120:
                    time.sleep(0.1)
                if store.add(key + "/noline", 0) > 0:
                    rec["ok"] = False
                    rec["stderr_tail"] = "rank 0's child exited 0 without a result line"
                    got = None
                    continue
                if li == 0:
                    first_mode = mi
                break
            log("rank %d: leg %s, mode %s failed (rc %s, %d rank(s))" % (rank, leg["name"], name, r["rc"], r["failed_ranks"]))
        if li == 0:
            if got is None:
                if rank == 0:
                    line = {"metric": "ADMM inner iterations/sec", "value": None, "unit": "iterations/s", "n_gpus": world,
                            "error": "no mode of the distributed loop produced a result", "attempts": attempts}
                return line, 1
            line = got
            if rank == 0:
                line["attempts"] = attempts
        elif rank == 0:
            if got is None:
                line[leg["name"]] = {"error": "leg failed in every mode tried", "attempts": attempts}
            else:
                got["attempts"] = attempts
                line[leg["name"]] = got
    return line, 0
