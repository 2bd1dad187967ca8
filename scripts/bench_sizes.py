#!/usr/bin/env python3
"""Single-GPU sweep over the BASELINE.json configs (size-matched synthetic grids unless the real
files are present): iterations/s, per-kernel HIP-event time of k_branch, solver work counters."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for case, steps in (("case1354pegase", 300), ("case13659pegase", 200), ("case_ACTIVSg70k", 100), ("synthetic1M", 30)):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--case", case, "--steps", str(steps), "--warmup", "20",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(case, "FAILED", r.stderr[-400:])
        continue
    d = json.loads(line[-1])
    print(json.dumps({"case": case, "iters_per_s": round(d["value"], 1), "ms_per_iter": round(d["ms_per_step"], 4),
                      "k_branch_ms": round(d["roofline"]["kernel_ms"], 4), "branch_GBps": round(d["roofline"]["achieved"], 1),
                      "iter_GBps": round(d["roofline"]["iteration_frac"] * 8000, 1), "avg_auglag": round(d["solver"]["avg_auglag_it"], 3),
                      "nline": d["config"]["nline"]}), flush=True)
