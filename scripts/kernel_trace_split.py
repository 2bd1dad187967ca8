#!/usr/bin/env python3
"""Split the rocprofv3 kernel trace of `bench.py --steps K --warmup W [--prep P]` by the phases of the run, so that the
trace's average kernel durations can be compared with the HIP-event figure of the bench line (which covers the timed
region only; the plain --stats average also covers the cold preparation iterations and the two whole solves).
usage: kernel_trace_split.py <kernel_trace.csv> [K=200] [W=20] [P=300] [trusted]  ->  JSON on stdout"""
import csv
import json
import sys

import numpy as np

f = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
P = int(sys.argv[4]) if len(sys.argv) > 4 else 300
SELFCHECK = (len(sys.argv) <= 5 or sys.argv[5] != "trusted")   # 5th argument `trusted`: the run had option trust_resident set (no self-check launches)
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
# resident loop: one k_resident launch per phase (preparation, warm-up, timed region, event-bracketed pass), `its` iterations each.
# Round 5: the library's first-use self check puts ONE more launch (8 iterations) of a kernel form behind the first call that used that form
# (exaadmm_resident_selfcheck_stats) -- the second launch of every distinct template instance in a process is that check, not a phase.
res_rows = [r for r in rows if "k_resident_t" in r["Kernel_Name"]]
if res_rows:
    form = lambda r: r["Kernel_Name"].split("(")[0]
    seen, phases, checks = {}, [], []
    for r in res_rows:
        k = form(r)
        seen[k] = seen.get(k, 0) + 1
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        (checks if (seen[k] == 2 and SELFCHECK) else phases).append((k, d))
    out["k_resident"] = {"launches": len(res_rows), "self_check_launches": [{"kernel": k.replace("void exa::", ""), "iterations": 8, "launch_us": float(d),
                                                                            "us_per_iteration": float(d / 8)} for k, d in checks]}
    for n, its, (k, d) in zip(["preparation", "warm_up", "timed_region", "event_bracketed_pass"], [P, W, K, K], phases):
        out["k_resident"][n] = {"kernel": k.replace("void exa::", ""), "iterations": its, "launch_us": float(d), "us_per_iteration": float(d / max(its, 1))}
print(json.dumps(out, indent=1))
