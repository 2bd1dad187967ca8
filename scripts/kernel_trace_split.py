#!/usr/bin/env python3
"""Split the rocprofv3 kernel trace of `bench.py --steps K --warmup W [--prep P]` by the phases of the run, so that the
trace's average kernel durations can be compared with the HIP-event figure of the bench line (which covers the timed
region only; the plain --stats average also covers the cold preparation iterations and the two whole solves).
usage: kernel_trace_split.py <kernel_trace.csv> [K=200] [W=20] [P=300] [solve|prep]  ->  JSON on stdout
(solve: the run timed a state on the solve's trajectory, bench.py --state solve, the default since round 6; prep: --state prep)"""
import csv
import json
import sys

import numpy as np

f = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
P = int(sys.argv[4]) if len(sys.argv) > 4 else 300
MODE = sys.argv[5] if len(sys.argv) > 5 else "solve"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = {"trace": f, "phases": "preparation %d | warm-up %d | timed region %d | event-bracketed pass %d | whole solves (rest)" % (P, W, K, K)}
for key, tag in (("k_branch", "k_branch"), ("k_bus", "k_bus_group")):
    d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]])
    if len(d) < P + W + K:   # (resident loop: the two-launch kernels only appear where a loop is completed; nothing to split by phase)
        if len(d):
            out[tag] = {"launches": int(len(d)), "average_us_all": float(d.mean())}
        continue
    cuts = [0, P, P + W, P + W + K, P + W + 2 * K, len(d)]
    names = ["preparation", "warm_up", "timed_region", "event_bracketed_pass", "whole_solves"]
    out[tag] = {"launches": int(len(d)), "average_us_all": float(d.mean())}
    for n, a, b in zip(names, cuts[:-1], cuts[1:]):
        if b > a:
            out[tag][n] = {"launches": int(b - a), "average_us": float(d[a:b].mean()), "min_us": float(d[a:b].min()), "max_us": float(d[a:b].max())}
# resident loop: a phase of `its` iterations is ceil(its / 1024) k_resident launches (a launch covers at most 1 024 iterations).  A fresh process
# runs, in this order: the preparation (plain kernel form when it starts an inner loop) | the first-use self check of that form (ONE resident
# launch of 8 iterations; its two-launch leg uses other kernels) | the warm-up (the heavy form where the grid leaves SIMDs idle and the loop is
# past its 64th iteration) | the self check of THAT form (8 iterations) | R repeats of the timed region | the event-bracketed pass.  The self
# check legs are told from the phases by their position in that order (round 5's version counted from the end and called the heavy self
# check "warm_up": VERDICT r05 "evidence bugs").
res_rows = [r for r in rows if "k_resident_t" in r["Kernel_Name"]]
if res_rows:
    nl = lambda its: max(1, (its + 1023) // 1024) if its > 0 else 0
    us = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = lambda r: r["Kernel_Name"].split("(")[0].replace("void exa::", "")
    is_heavy = lambda r: "k_resident_t<1, true" in r["Kernel_Name"]
    out["k_resident"] = {"launches": len(res_rows), "launches_per_region": nl(K)}
    pos = 0
    seen_forms = set()

    def take(n):
        global pos
        grp = res_rows[pos:pos + n]
        pos += len(grp)
        return grp

    def phase(label, its, grp):
        if not grp:
            return
        d = sum(us(r) for r in grp)
        span = (int(grp[-1]["End_Timestamp"]) - int(grp[0]["Start_Timestamp"])) / 1e3
        out["k_resident"][label] = {"kernel": name(grp[-1]), "iterations": its, "launches": len(grp), "launch_us": float(d),
                                    "us_per_iteration": float(d / max(its, 1)), "first_start_to_last_end_us_per_iteration": float(span / max(its, 1))}

    def self_check(label, form_rows):
        # the call that first used a kernel form is followed by ONE resident launch of 8 iterations of that form
        form = name(form_rows[-1])
        if form in seen_forms:
            return
        seen_forms.add(form)
        if pos < len(res_rows) and name(res_rows[pos]) == form:
            phase(label, 8, take(1))

    if MODE == "solve":
        # bench.py --state solve --repeats 1 --no-deep-loop-leg (and no other legs): the run ENDS with warm-up | timed region | event-bracketed pass;
        # everything in front of them brings the state about -- the inner loops of the outer iterations before the timed one (two launches each
        # where the first one is split for the heavy wavefronts), the first-use self checks of both kernel forms (they follow the first call that
        # used a form: the first inner loop), the untimed iterations of the timed outer iteration
        tail = nl(W) + 2 * nl(K)
        head = res_rows[:max(0, len(res_rows) - tail)]
        out["k_resident"]["state_preparation"] = {"launches": len(head), "launch_us": float(sum(us(r) for r in head)),
                                                  "note": "inner loops of the earlier outer iterations, first-use self checks (8 iterations per kernel form), the untimed iterations of the timed outer iteration"}
        pos = len(head)
        phase("warm_up", W, take(nl(W)))
        phase("timed_region", K, take(nl(K)))
        phase("event_bracketed_pass", K, take(nl(K)))
        print(json.dumps(out, indent=1))
        sys.exit(0)
    grp = take(nl(P))
    phase("preparation", P, grp)
    if grp:
        self_check("self_check_after_preparation", grp)
    grp = take(nl(W))
    phase("warm_up", W, grp)
    if grp:
        self_check("self_check_after_warm_up", grp)
    rest = res_rows[pos:]
    per = nl(K)
    # what is left: R repeats of the timed region (the FIRST may be followed by a self check when it is the first call of its form), then
    # the event-bracketed pass
    if rest and name(rest[0]) not in seen_forms and len(rest) > per and (len(rest) - 1) % per == 0:
        first = take(per)
        phase("timed_region_first", K, first)
        self_check("self_check_after_first_region", first)
        rest = res_rows[pos:]
    if rest and len(rest) % per == 0:
        R = len(rest) // per - 1
        reps = [rest[i * per:(i + 1) * per] for i in range(R)]
        if reps:
            durs = sorted(sum(us(r) for r in g) for g in reps)
            med = [g for g in reps if sum(us(r) for r in g) == durs[len(durs) // 2]][0]
            phase("timed_region", K, med)
            out["k_resident"]["timed_region"].update({"repeats": R, "which": "the median repeat", "us_per_iteration_min": durs[0] / K, "us_per_iteration_max": durs[-1] / K})
        phase("event_bracketed_pass", K, rest[R * per:])
    else:
        out["k_resident"]["unclassified"] = {"launches": len(rest), "note": "launch count does not fit R x %d + %d" % (per, per)}
print(json.dumps(out, indent=1))
