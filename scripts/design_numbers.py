#!/usr/bin/env python3
"""Regenerates the table of current numbers in DESIGN.md (between the markers `<!-- numbers:begin -->` / `<!-- numbers:end -->`) from the
bench lines under profiles/ -- so that the document quotes what the committed JSON says and nothing else -- and the figures README.md quotes in its
"Benchmark" paragraph (same source).
    python scripts/design_numbers.py [round]          (default r06; reads profiles/<round>_bench*.json, <round>_pmc_summary.json)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(ROOT, "profiles")


def load(name):
    f = os.path.join(P, "%s_%s.json" % (rnd, name))
    return json.load(open(f)) if os.path.exists(f) else None


def k(x, nd=0):
    return "—" if x is None else (("%%.%df" % nd) % x if nd else "{:,.0f}".format(x).replace(",", " "))


rows = []
for label, name in (("case1354pegase", "bench_case1354pegase"), ("**case13659pegase** (headline; default command)", "bench"),
                    ("case13659pegase, the driver's command (K = 20, W = 5)", "bench_driver_command_steps20"),
                    ("ACTIVSg70k", "bench_case_ACTIVSg70k"), ("synthetic 1M branches", "bench_synthetic1M")):
    d = load(name)
    if not d:
        continue
    r, fs = d["roofline"], d.get("full_solve") or {}
    cp = (r.get("critical_path") or {})
    form = r.get("resident_form") or {}
    loop = ("resident, %d tile(s) per wavefront%s" % (form["tiles"], ", %d heavy wavefronts" % form["heavy"] if form.get("heavy") else "")) if form.get("tiles") else "two launches per iteration"
    deep = (d.get("deep_loop_state") or {}).get("value")
    if deep is None and "preparation iterations" in d["config"].get("state", ""):   # (a line timed with --state prep: its value IS the deep-loop figure)
        deep = d["value"]
    on_traj = "—" if "preparation iterations" in d["config"].get("state", "") else "%s (%.4f ms)" % (k(d["value"]), d["ms_per_step"])
    rows.append("| %s | %s | %s (%s / %s) | %s | %s | %s | %s | %.3f | %s | %s |" % (
        label, k(d["config"]["nline"]), k(fs.get("iterations_per_s")), fs.get("outer", "—"), fs.get("cumul", "—"),
        k((fs.get("cpu_port") or {}).get("iterations_per_s"), 1), on_traj, k(deep), loop,
        r["frac"], ("%.2f" % cp["frac"]) if cp.get("frac") else "—", k(r.get("traffic"))))
out = ["| grid (synthetic, size-matched) | branches | whole solve it/s (outer / inner iterations) | CPU port, same solve, 1 core (it/s) | `value`: from a state on the solve's trajectory, it/s (ms per iteration) | deep-loop state it/s (rounds 1-5's `value`) | loop of the timed regions | roofline.frac (HBM) of `value` | critical_path.frac | PMC traffic per iteration (B) |",
       "|---|---|---|---|---|---|---|---|---|---|"] + rows
extra = []
d = load("bench")
if d and d.get("north_star_scaling") and "value" in d["north_star_scaling"]:
    ns = d["north_star_scaling"]
    extra.append("`north_star_scaling` of the headline line (one area of ACTIVSg70k's size on one GPU): %s it/s, %s." % (k(ns["value"]), ns["config"].get("resident_form")))
if d and d.get("throughput_1M") and "value" in d["throughput_1M"]:
    tp = d["throughput_1M"]
    extra.append("`throughput_1M` of the headline line (BASELINE.json configs[4] on one GPU, 20-iteration regions): %s it/s, k_branch %.4f ms, roofline.frac %.3f." % (k(tp["value"]), tp["roofline"]["kernel_ms"], tp["roofline"]["frac"]))
d = load("bench_force_dist_one_rank")
if d:
    extra.append("`bench.py --force-dist` (the N > 1 code path with one rank, supervised child): %s it/s; attempts %s." % (k(d["value"]), [a["mode"] for a in d.get("attempts", [])]))
d = load("bench_mp_4_periods")
if d:
    extra.append("Multi-period model, 4 periods on the headline grid: %s it/s (%.4f ms per iteration)." % (k(d["value"]), d["ms_per_step"]))
for nm, lab in (("bench_case9_real_file", "case9 (real file)"), ("bench_case118_real_file", "case118 (real file)")):
    d = load(nm)
    if d:
        fs = d.get("full_solve") or {}
        extra.append("%s: steady state %s it/s, whole solve %s it/s (%s / %s iterations, objective %s)." % (lab, k(d["value"]), k(fs.get("iterations_per_s")), fs.get("outer"), fs.get("cumul"), fs.get("objval")))
block = "\n".join(out) + "\n\n" + "\n".join("* " + e for e in extra) + "\n"
path = os.path.join(ROOT, "DESIGN.md")
txt = open(path).read()
new = re.sub(r"(<!-- numbers:begin -->\n).*?(<!-- numbers:end -->)", lambda m: m.group(1) + block + m.group(2), txt, flags=re.S)
open(path, "w").write(new)
print(block)


# README.md: the figures of its benchmark paragraph, from the same lines
b, dcmd, mp = load("bench"), load("bench_driver_command_steps20"), load("bench_mp_4_periods")
if b and dcmd and mp:
    fs, cb, deep = b["full_solve"], b["cpu_baseline"], b["deep_loop_state"]["value"]
    subs = [(r"\d\d \d{3}\n  inner iterations/s with the default", k(b["value"]) + "\n  inner iterations/s with the default"),
            (r"\d\d \d{3} with the driver", k(dcmd["value"]) + " with the driver"),
            (r"\d\d \d{3} it/s \(≈ \d+× the serial CPU port\)", "%s it/s (≈ %d× the serial CPU port)" % (k(deep), round(deep / cb["value"]))),
            (r"\d+ \d{3} it/s, ≈ \d+× the CPU port on the same solve", "%s it/s, ≈ %d× the CPU port on the same solve" % (k(fs["iterations_per_s"]), round(fs["speedup_vs_cpu_port"]))),
            (r"\(88 207 branches\): \d \d{3} it/s", "(88 207 branches): %s it/s" % k(b["north_star_scaling"]["value"])),
            (r"configs\[4\]: \d \d{3} it/s, `k_branch` at 0\.\d\d", "configs[4]: %s it/s, `k_branch` at %.2f" % (k(b["throughput_1M"]["value"]), b["throughput_1M"]["roofline"]["frac"])),
            (r"the multi-period model, \d\d \d{3} it/s", "the multi-period model, %s it/s" % k(mp["value"]))]
    rp = os.path.join(ROOT, "README.md")
    rt = open(rp).read()
    for pat, new_text in subs:
        assert len(re.findall(pat, rt)) == 1, pat
        rt = re.sub(pat, lambda m, t=new_text: t, rt)
    open(rp, "w").write(rt)
