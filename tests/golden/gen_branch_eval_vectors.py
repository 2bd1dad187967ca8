#!/usr/bin/env python3
"""Generate tests/golden/branch_eval_vectors.json by RUNNING the reference's own
branch callbacks (objective / gradient / Hessian of the branch augmented
Lagrangian) on seeded random inputs.

The reference is Julia and no Julia runtime exists in this container, but the
three callbacks in
    /root/reference/src/models/acopf/acopf_eval_linelimit_kernel_cpu.jl
      eval_f_polar_linelimit_kernel_cpu      (:1-46)
      eval_grad_f_polar_linelimit_kernel_cpu (:48-172)
      eval_h_polar_linelimit_kernel_cpu      (:174-605, ":Values" branch)
are straight-line scalar arithmetic, so this script reads that file AT
GENERATION TIME, rewrites Julia surface syntax to Python token-by-token
(1-based -> 0-based literal indices, ^ -> **, `param[k,I]` -> `param[k-1]`)
and executes the statements.  No reference text is stored in this repository:
only the numeric inputs/outputs (data) are committed.  Must be run in the
build container (needs /root/reference); the tests only read the JSON.
"""
import json
import math
import os
import re
import sys

import numpy as np

REF = "/root/reference/src/models/acopf/acopf_eval_linelimit_kernel_cpu.jl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "branch_eval_vectors.json")


def _to_python(lines):
    out = []
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith("#"):
            continue
        s = s.split("#")[0].rstrip()
        s = re.sub(r"param\[(\d+),I\]", lambda m: "param[%d]" % (int(m.group(1)) - 1), s)
        s = re.sub(r"\bx\[(\d+)\]", lambda m: "x[%d]" % (int(m.group(1)) - 1), s)
        s = re.sub(r"\bg\[(\d+)\]", lambda m: "g[%d]" % (int(m.group(1)) - 1), s)
        s = s.replace("values[nz]", "values[nz-1]")
        s = s.replace("^", "**")
        out.append(s)
    return "\n".join(out)


def _function_bodies():
    src = open(REF).read().splitlines()
    starts = [i for i, l in enumerate(src) if l.startswith("function ")]
    assert len(starts) == 3, starts
    starts.append(len(src))
    bodies = []
    for a, b in zip(starts[:-1], starts[1:]):
        chunk = src[a:b]
        i0 = next(i for i, l in enumerate(chunk) if "@inbounds begin" in l) + 1
        bodies.append(chunk[i0:])
    # f: statements up to "f *= scale"
    f_lines = []
    for l in bodies[0]:
        if l.strip() == "end":
            continue
        f_lines.append(l)
        if l.strip().startswith("f *= scale"):
            break
    # grad: up to g[6] = ...
    g_lines = []
    for l in bodies[1]:
        g_lines.append(l)
        if l.strip().startswith("g[6]"):
            break
    # hess: the ":Values" branch: from the "else" after the structure block
    h = bodies[2]
    i_else = next(i for i, l in enumerate(h) if l.strip() == "else")
    h_lines = []
    for l in h[i_else + 1:]:
        if l.strip() in ("end",):
            break
        h_lines.append(l)
    return _to_python(f_lines), _to_python(g_lines), _to_python(h_lines)


# (row, col) order of the 20 values, as listed in the ":Structure" branch (:183-207)
HESS_ORDER = [(1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (2, 2), (3, 2), (4, 2), (5, 2), (6, 2),
              (3, 3), (4, 3), (5, 3), (6, 3), (4, 4), (5, 4), (6, 4), (5, 5), (6, 6)]


def main():
    f_src, g_src, h_src = _function_bodies()
    rng = np.random.default_rng(20261002)
    cases = []
    for _ in range(24):
        Y = rng.uniform(-12, 12, 8)
        x = np.array([rng.uniform(0.9, 1.1), rng.uniform(0.9, 1.1), rng.uniform(-0.6, 0.6),
                      rng.uniform(-0.6, 0.6), rng.uniform(-3, 0), rng.uniform(-3, 0)])
        param = np.zeros(31)
        param[0:8] = rng.uniform(-50, 50, 8)
        param[8:12] = rng.choice([10.0, 50.0, 400.0, 3e4])
        param[12:16] = rng.choice([1e3, 5e3, 4e4, 3e5])
        param[16:24] = rng.uniform(-2, 2, 8)
        param[24:26] = rng.uniform(-5, 5, 2)
        param[26] = 10.0 ** rng.integers(1, 6)
        scale = float(rng.choice([1e-4, 1e-5, 1.0]))
        ns = dict(cos=math.cos, sin=math.sin, x=list(x), param=list(param), scale=scale, f=0.0,
                  YffR=Y[0], YffI=Y[1], YftR=Y[2], YftI=Y[3], YttR=Y[4], YttI=Y[5], YtfR=Y[6], YtfI=Y[7])
        ns["f"] = 0.0
        exec(f_src, ns)
        fval = ns["f"]
        ns["g"] = [0.0] * 6
        exec(g_src, ns)
        ns["values"] = [0.0] * 20
        exec(h_src, ns)
        H = np.zeros((6, 6))
        for (r, c), v in zip(HESS_ORDER, ns["values"]):
            H[r - 1, c - 1] = v
            H[c - 1, r - 1] = v
        cases.append(dict(Y=Y.tolist(), x=x.tolist(), param=param.tolist(), scale=scale,
                          f=fval, g=list(ns["g"]), H=H.tolist()))
    json.dump(dict(source="acopf_eval_linelimit_kernel_cpu.jl executed via syntax rewrite",
                   cases=cases), open(OUT, "w"), indent=0)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference tree not present; fixture can only be regenerated in the build container")
    main()
