#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (idle queue time between the end of one
dispatch and the start of the next), per kernel kind.  usage: kernel_gaps.py kernel_trace.csv [last N kernels]"""
import collections
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) - 1
rows = rows[-n - 1:]


def kind(r):
    k = r["Kernel_Name"]
    for s in ("k_branch_t", "k_bus_group", "k_fused_zl", "k_final_reduce", "k_sumsq_z", "k_update_lz", "copyBuffer", "fillBuffer"):
        if s in k:
            return s
    return k[:30]


st = np.array([int(r["Start_Timestamp"]) for r in rows])
en = np.array([int(r["End_Timestamp"]) for r in rows])
names = [kind(r) for r in rows]
dur, gap = en - st, st[1:] - en[:-1]
print("%d kernels, span %.3f ms: busy %.3f ms, idle between kernels %.3f ms" % (len(rows), (en[-1] - st[0]) / 1e6, dur.sum() / 1e6, gap.sum() / 1e6))
print("%-16s %6s %10s | gap BEFORE the kernel, by predecessor: mean us (count)" % ("kernel", "n", "dur us"))
for key in sorted(set(names), key=lambda q: -sum(1 for x in names if x == q)):
    m = np.array([x == key for x in names])
    by = collections.defaultdict(list)
    for i in np.flatnonzero(m[1:]) + 1:
        by[names[i - 1]].append(gap[i - 1] / 1e3)
    txt = ", ".join("%s %.2f (%d)" % (p, np.mean(v), len(v)) for p, v in sorted(by.items(), key=lambda kv: -len(kv[1])))
    print("%-16s %6d %10.2f | %s" % (key, m.sum(), dur[m].mean() / 1e3, txt))
