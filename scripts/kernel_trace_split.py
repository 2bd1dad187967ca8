#!/usr/bin/env python3
"""Split the rocprofv3 kernel trace of `bench.py --steps K --warmup W [--prep P]` by the phases of the run, so that the
trace's average kernel durations can be compared with the HIP-event figure of the bench line (which covers the timed
region only; the plain --stats average also covers the cold preparation iterations and the two whole solves).
usage: kernel_trace_split.py <kernel_trace.csv> [K=200] [W=20] [P=300] [chunk=0]  ->  JSON on stdout"""
import csv
import json
import sys

import numpy as np

f = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
P = int(sys.argv[4]) if len(sys.argv) > 4 else 300
CHUNK = int(sys.argv[5]) if len(sys.argv) > 5 else int(__import__('os').environ.get('EXAADMM_RESIDENT_CHUNK', '0') or 0)   # option resident_chunk of the run (0, the default: one launch per region)
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
# resident loop: one k_resident launch per phase (preparation, warm-up, timed region, event-bracketed pass), `its` iterations each -- or, on a grid
# with heavy wavefronts, ceil(its / CHUNK) launches per phase (option resident_chunk, 50: every launch chooses its heavy wavefronts anew).  The
# phases are counted FROM THE END of the trace (the event-bracketed pass is the last thing a --no-full-solve run does): the launches in front of
# the warm-up are the preparation and the first-use self checks (8 iterations each, one per kernel form).
res_rows = [r for r in rows if "k_resident_t" in r["Kernel_Name"]]
if res_rows:
    heavy = any("k_resident_t<1, true" in r["Kernel_Name"] for r in res_rows)
    chunk = CHUNK if (heavy and CHUNK > 0) else 0
    nl = lambda its: (its + chunk - 1) // chunk if chunk else 1
    us = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out["k_resident"] = {"launches": len(res_rows), "launches_per_region": nl(K), "chunk": chunk}
    end = len(res_rows)
    for n, its in (("event_bracketed_pass", K), ("timed_region", K), ("warm_up", W)):
        grp = res_rows[end - nl(its):end]
        end -= nl(its)
        d = sum(us(r) for r in grp)
        span = (int(grp[-1]["End_Timestamp"]) - int(grp[0]["Start_Timestamp"])) / 1e3
        out["k_resident"][n] = {"kernel": grp[-1]["Kernel_Name"].split("(")[0].replace("void exa::", ""), "iterations": its, "launches": len(grp), "launch_us": float(d),
                                "us_per_iteration": float(d / max(its, 1)), "first_start_to_last_end_us_per_iteration": float(span / max(its, 1))}
    rest = res_rows[:end]
    out["k_resident"]["preparation_and_self_checks"] = {"launches": len(rest), "launch_us": float(sum(us(r) for r in rest)),
                                                        "note": "%d preparation iterations (+ 8 per first-use self check of a kernel form, resident leg)" % P}
print(json.dumps(out, indent=1))
