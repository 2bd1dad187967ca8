#!/usr/bin/env python3
"""Copy the artefacts of a `final` GPU batch (bench lines, rocprofv3 kernel stats, PMC passes) from gpurun_out/<dir>
into profiles/ under the round's names.  usage: refresh_profiles.py gpurun_out/r02_final3 r02 "<build description>" """
import glob
import json
import os
import shutil
import subprocess
import sys

src, rnd, build = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(f):
    for l in open(f):
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit("no JSON line in " + f)


names = {"bench_default": "bench", "bench_driver_cmd": "bench_driver_command_steps20", "bench_case1354pegase": "bench_case1354pegase",
         "bench_case_ACTIVSg70k": "bench_case_ACTIVSg70k", "bench_synthetic1M": "bench_synthetic1M", "bench_dist1": "bench_force_dist_one_rank",
         "bench_mp4": "bench_mp_4_periods", "bench_case118": "bench_case118_real_file", "bench_case9": "bench_case9_real_file"}
for k, v in names.items():
    f = os.path.join(src, k + ".log")
    if os.path.exists(f):
        json.dump(line(f), open(os.path.join(ROOT, "profiles", "%s_%s.json" % (rnd, v)), "w"), indent=1)
st = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % rnd))
out_p = os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % rnd)
old = json.load(open(out_p)) if os.path.exists(out_p) else {}
out = {"_note": old.get("_note", ""), "build": build}
if "previous_build_r02_a" in old:
    out["previous_build_r02_a"] = old["previous_build_r02_a"]
for d, wl, steps, G, L in (("pmc_case13659pegase", "synthetic-case13659pegase-sized", 200, 4092, 20467), ("pmc_synthetic1M", "synthetic-synthetic1M-sized", 50, 50000, 1000000)):
    p = os.path.join(src, d)
    if not os.path.isdir(p):
        continue
    out.update(json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "summarize_pmc.py"), p, wl, str(steps)])))
    fused = True   # the z/l pass rides in the branch launch at every size (since the load-first prologue)
    out[wl]["k_branch"]["algorithmic_bytes"] = int(824.0 * L + 192.0 * G) if fused else int(568.0 * L)
    out[wl]["k_branch"]["algorithmic_bytes_note"] = ("fused launch: 568 B per branch + 128 B per generator + the unshared part of the z/l pass (32 B per vector element) = 824 L + 192 G"
                                                     if fused else "568 B per branch")
json.dump(out, open(out_p, "w"), indent=1)
print("profiles refreshed from", src)
