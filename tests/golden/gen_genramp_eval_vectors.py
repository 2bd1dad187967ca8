#!/usr/bin/env python3
"""Generate tests/golden/genramp_eval_vectors.json by RUNNING the reference's own generator
ramp callbacks (multi-period ACOPF)
    /root/reference/src/models/mpacopf/mpacopf_eval_generator_kernel_cpu.jl
      eval_f_generator_kernel_cpu       (:1-27)
      eval_grad_f_generator_kernel_cpu  (:29-46)
      eval_h_generator_kernel_cpu       (:48-101, value branch)
on seeded random inputs.  Same method as gen_branch_eval_vectors.py: the file is read AT
GENERATION TIME, its straight-line arithmetic is rewritten from Julia to Python surface syntax
(1-based -> 0-based literal indices, ^ -> **) and executed; only numbers are committed.
Run in the build container (needs /root/reference)."""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference/src/models/mpacopf/mpacopf_eval_generator_kernel_cpu.jl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "genramp_eval_vectors.json")


def _to_python(lines):
    out = []
    for ln in lines:
        s = ln.split("#")[0].strip()
        if not s:
            continue
        s = re.sub(r"param\[(\d+),I\]", lambda m: "param[%d]" % (int(m.group(1)) - 1), s)
        s = re.sub(r"\bx\[(\d+)\]", lambda m: "x[%d]" % (int(m.group(1)) - 1), s)
        s = re.sub(r"\bg\[(\d+)\]", lambda m: "g[%d]" % (int(m.group(1)) - 1), s)
        s = s.replace("values[nz]", "values[nz-1]").replace("^", "**")
        out.append(s)
    return "\n".join(out)


def _bodies():
    src = open(REF).read().splitlines()
    starts = [i for i, l in enumerate(src) if l.startswith("function ")]
    f0, g0, h0 = starts[0], starts[1], starts[2]      # the first three functions: non-"loose" model
    def body(a, b):
        chunk = src[a:b]
        i0 = next(i for i, l in enumerate(chunk) if "@inbounds begin" in l) + 1
        return chunk[i0:]
    f_lines = [l for l in body(f0, g0) if l.strip().startswith("f +=")]
    g_lines = [l for l in body(g0, h0) if re.match(r"\s*g\[\d\]", l)]
    h = body(h0, starts[3])
    i_else = next(i for i, l in enumerate(h) if l.strip() == "else")
    h_lines = [l for l in h[i_else + 1:] if l.strip().startswith(("nz", "values"))]
    return _to_python(f_lines), _to_python(g_lines), _to_python(h_lines)


HESS_ORDER = [(1, 1), (2, 1), (3, 1), (2, 2), (3, 2), (3, 3)]   # column-major lower triangle (:58-66)


def main():
    f_src, g_src, h_src = _bodies()
    rng = np.random.default_rng(20261003)
    cases = []
    for _ in range(24):
        x = np.array([rng.uniform(0, 4), rng.uniform(0, 4), rng.uniform(-0.1, 0.1)])
        param = np.array([rng.normal(0, 20), rng.normal(0, 20), rng.choice([10.0, 50.0, 400.0]), rng.choice([10.0, 50.0, 400.0]),
                          rng.uniform(0, 4), rng.uniform(0, 4), rng.normal(0, 30), 10.0 ** rng.integers(1, 7)])
        c2, c1, c0, M = rng.uniform(0, 0.2), rng.uniform(1, 40), rng.uniform(0, 600), 100.0
        scale = float(rng.choice([1.0, 1e-4]))
        ns = dict(x=list(x), param=list(param), scale=scale, c2=c2, c1=c1, c0=c0, baseMVA=M, f=0.0)
        exec(f_src, ns)
        fval = ns["f"] * scale                     # the function ends with f *= scale
        ns["g"] = [0.0] * 3
        exec(g_src, ns)
        ns["values"] = [0.0] * 6
        ns["nz"] = 0
        exec(h_src, ns)
        H = np.zeros((3, 3))
        for (r, c), v in zip(HESS_ORDER, ns["values"]):
            H[r - 1, c - 1] = H[c - 1, r - 1] = v
        cases.append(dict(x=x.tolist(), param=param.tolist(), scale=scale, c2=c2, c1=c1, c0=c0, baseMVA=M,
                          f=fval, g=list(ns["g"]), H=H.tolist()))
    json.dump(dict(source="mpacopf_eval_generator_kernel_cpu.jl executed via syntax rewrite", cases=cases), open(OUT, "w"), indent=0)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference tree not present; fixture can only be regenerated in the build container")
    main()
