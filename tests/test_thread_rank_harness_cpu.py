"""CPU: the retry rule of the thread-rank GPU tests (tests/test_gpu_dist_threads.py _run_worker).  Ranks that are threads of one process on one GPU can
have a kernel held up behind another rank's (profiles/r06_heavy_split_on_partitions.txt); since round 6 that ends LOUDLY -- a failed launch with its
message, or a failed call -- and such a run of the mailbox loop is repeated once.  What must never happen: a retry after a wrong number, a retry of the
loops that have no kernels waiting for each other, a second retry."""
import json
import types

import pytest

import test_gpu_dist_threads as T


class _Runs:
    def __init__(self, outcomes):
        self.outcomes, self.calls = list(outcomes), 0

    def __call__(self, *a, **k):
        self.calls += 1
        rc, out, err = self.outcomes.pop(0)
        return types.SimpleNamespace(returncode=rc, stdout=out, stderr=err)


GOOD = "RESULT " + json.dumps({"cumul": [1232]})
WRONG = "RESULT " + json.dumps({"cumul": [1938]})
LOUD_CALL = "AssertionError: ['ExaAdmmError(\"exaadmm_norm_z: rank 0 of 3: a peer did not answer a sums exchange through the mailboxes within 3 s\")']"
LOUD_LAUNCH = "RESULT " + json.dumps({"cumul": [1232], "dist_stats": [{"downgrades": 1, "last_error": "resident launch did not complete on device 0 (wait code 8"}]})


@pytest.fixture
def harness(monkeypatch):
    monkeypatch.setattr(T, "_shim", lambda: "/nowhere/librccl_shim.so")
    for var in ("EXAADMM_NO_SPEC", "EXAADMM_BUS_KERNEL"):
        monkeypatch.delenv(var, raising=False)

    def install(outcomes):
        runs = _Runs(outcomes)
        monkeypatch.setattr(T.subprocess, "run", runs)
        return runs
    return install


def test_a_loud_failure_of_the_mailbox_loop_is_repeated_once(harness):
    runs = harness([(1, "", LOUD_CALL), (0, GOOD, "")])
    assert T._run_worker(3, "case", EXAADMM_DIST_P2P="1")["cumul"] == [1232] and runs.calls == 2
    runs = harness([(0, LOUD_LAUNCH, ""), (0, GOOD, "")])       # (a launch that failed and was repeated with loop B: rc 0, but the test asserts "no downgrade")
    assert T._run_worker(3, "case", EXAADMM_DIST_P2P="1")["cumul"] == [1232] and runs.calls == 2


def test_a_wrong_number_is_never_repeated(harness):
    runs = harness([(0, WRONG, ""), (0, GOOD, "")])
    assert T._run_worker(3, "case", EXAADMM_DIST_P2P="1")["cumul"] == [1938] and runs.calls == 1   # (the caller's asserts see it)
    runs = harness([(1, "", "AssertionError: a rank hung"), (0, GOOD, "")])
    with pytest.raises(AssertionError):
        T._run_worker(3, "case", EXAADMM_DIST_P2P="1")
    assert runs.calls == 1


def test_there_is_no_second_retry_and_none_for_the_loops_without_waiting_kernels(harness):
    runs = harness([(1, "", LOUD_CALL), (1, "", LOUD_CALL), (0, GOOD, "")])
    with pytest.raises(AssertionError):
        T._run_worker(3, "case", EXAADMM_DIST_P2P="1")
    assert runs.calls == 2
    runs = harness([(1, "", LOUD_CALL), (0, GOOD, "")])
    with pytest.raises(AssertionError):
        T._run_worker(3, "case")            # (EXAADMM_DIST_P2P=0: two launches per iteration + the stand-in's exchange)
    assert runs.calls == 1


def test_the_mailbox_tests_run_with_three_seconds_of_patience(harness, monkeypatch):
    seen = {}

    def run(cmd, env=None, **k):
        seen.update(env)
        return types.SimpleNamespace(returncode=0, stdout=GOOD, stderr="")
    monkeypatch.setattr(T.subprocess, "run", run)
    T._run_worker(2, "", EXAADMM_DIST_P2P="1")
    assert seen["EXAADMM_TEST_P2P_PATIENT_MS"] == "3000"
    seen.clear()
    T._run_worker(2, "")
    assert "EXAADMM_TEST_P2P_PATIENT_MS" not in seen
