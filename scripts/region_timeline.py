#!/usr/bin/env python3
"""What happens on the GPU between two resident launches of bench.py's timed regions?  Reads the kernel / memory-copy traces of
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 bench.py --steps 20 --warmup 5 --repeats 30 ...
and prints every event from the end of one k_resident launch to the end of the next (a region in the middle of the run).
    python scripts/region_timeline.py DIR"""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
res = [k for k, e in enumerate(ev) if "k_resident_t" in e[2]]
a, b = res[len(res) // 2], res[len(res) // 2 + 1]
t0 = ev[a][1]
print("from the end of one resident launch to the end of the next: %.1f us, the launch itself %.1f us" % ((ev[b][1] - t0) / 1e3, (ev[b][1] - ev[b][0]) / 1e3))
for s, e, n in ev[a + 1:b + 1]:
    print("  +%8.1f us  %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
