#!/usr/bin/env python3
"""Per-kernel, per-launch averages of rocprofv3 --pmc passes (scripts/collect_pmc.sh) over the TIMED region of
bench.py (with --repeats 1: the last 2 x `steps` launches of each kernel are the timed region and the event-bracketed
pass that follows it; the last `steps` are used), with the gfx950 corrections of MI355X_MICROARCH.md:
FETCH_SIZE is reported in KB and counts REQUESTS in units of 64 B (profiles/archive_r03_r04/r03_fetch_size_counter_check.txt): a coalesced
128-B request counts as 64 B -> read bytes = 2 x FETCH_SIZE x 1024 for plane-contiguous readers (branch kernel, elementwise
passes); requests of one 64-B sector or less (gathers, the 64-B bus records) are counted at their true size, so for the
bus kernel both bounds are reported: traffic_bytes (x 2, upper bound) and traffic_bytes_sector_reads (x 1);
WRITE_SIZE (KB) is exact for streaming stores; SQ_* cycle counters are in quad-cycles summed over waves;
GRBM_GUI_ACTIVE is summed over the 8 XCDs -> shader clock = GRBM_GUI_ACTIVE / 8 / kernel wall time.
usage: summarize_pmc.py OUTDIR workload-name [steps] -> JSON on stdout"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, workload = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
chunk = 1024 if steps > 1024 else 0   # a resident launch covers at most 1 024 iterations: a longer region is ceil(steps / 1024) launches
short = {"k_branch_t": "k_branch", "k_bus_group": "k_bus_group", "k_fused_zl": "k_fused_zl", "k_resident_t": "k_resident"}
# k_resident (resident loop): ONE launch does the `steps` iterations of a region; its last launch is the event-bracketed pass;
# counters and duration are divided by `steps` -> per ITERATION figures
acc = defaultdict(dict)
dur = {}
heavy_form = {}
for f in glob.glob(os.path.join(out_dir, "*", "*", "*counter_collection.csv")):
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "k_resident_t<1, true" in r["Kernel_Name"]:
            heavy_form[f] = True
        name = next((v for k, v in short.items() if k in r["Kernel_Name"]), None)
        if name is None:
            continue
        per[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for name, cs in per.items():
        for cname, vals in cs.items():
            vals.sort()
            if name == "k_resident":
                nl = (steps + chunk - 1) // chunk if chunk > 0 else 1
                tail = vals[-nl:]
                acc[name][cname] = sum(v for _, v, _ in tail) / steps
                dur.setdefault(name, {})[cname] = sum(d for _, _, d in tail) / steps * 1e-3
                continue
            tail = vals[-steps:]
            acc[name][cname] = sum(v for _, v, _ in tail) / len(tail)
            dur.setdefault(name, {})[cname] = sum(d for _, _, d in tail) / len(tail) * 1e-3   # us, under the profiler
res = {}
for name, c in acc.items():
    e = {k: round(v, 3) for k, v in c.items()}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["traffic_bytes"] = int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
        e["read_factor_applied"] = 2
        if name == "k_bus_group":
            e["traffic_bytes_sector_reads"] = int(c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
        e["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4)      # both in quad-cycles
    if "SQ_INSTS_VALU" in c and "SQ_WAVE_CYCLES" in c:
        e["cycles_per_valu_instruction"] = round(4 * c["SQ_WAVE_CYCLES"] / c["SQ_INSTS_VALU"], 3)
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        e["issue_stall_frac"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    if "GRBM_GUI_ACTIVE" in c:
        us = dur[name]["GRBM_GUI_ACTIVE"]
        e["kernel_us_in_that_pass"] = round(us, 2)
        e["shader_clock_mhz"] = round(c["GRBM_GUI_ACTIVE"] / 8 / us, 1)
    if name == "k_resident":
        e["per"] = "iteration (a region = %d iterations)" % steps
    res[name] = e
print(json.dumps({workload: res}, indent=1))
