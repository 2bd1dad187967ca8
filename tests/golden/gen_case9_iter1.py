#!/usr/bin/env python3
"""Extract the known-answer vectors of the reference's own ACOPF test
(/root/reference/test/algorithms/acopf_update_cpu.jl:28-151 and the full-solve
anchors :168-179) into tests/golden/case9_iter1.json.  Run in the build
container only (needs /root/reference); the tests read the JSON."""
import json, os, re, sys

REF = "/root/reference/test/algorithms/acopf_update_cpu.jl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "case9_iter1.json")

def main():
    src = open(REF).read()
    out = {"source": "test/algorithms/acopf_update_cpu.jl", "atol": 1e-6,
           "settings": {"rho_pq": 4e2, "rho_va": 4e4, "scale": 1e-4, "initial_beta": 1e3}}
    for name in ("U_GEN", "U_BR", "V_GEN", "V_BR", "Z_GEN", "Z_BR", "L_GEN", "L_BR"):
        m = re.search(r"\b%s\s*=\s*\[(.*?)\]" % name, src, re.S)
        out[name] = [float(t) for t in re.findall(r"-?\d+\.?\d*(?:[eE][-+]?\d+)?", m.group(1))]
    assert len(out["U_GEN"]) == 6 and len(out["U_BR"]) == 72
    m = re.search(r"solve_acopf\(case; outer_iterlim=(\d+), rho_pq=rho_pq, rho_va=rho_va, outer_eps=2\*1e-5", src)
    out["solve_case9"] = {"outer_iterlim": int(m.group(1)), "outer_eps": 2e-5,
                          "status": "Solved",
                          "outer": int(re.search(r"mod\.info\.outer == (\d+)", src).group(1)),
                          "cumul": int(re.search(r"mod\.info\.cumul == (\d+)", src).group(1)),
                          "objval": float(re.search(r"objval, ([\d.]+); atol=1e-3", src).group(1)),
                          "objval_atol": 1e-3}
    cum = re.findall(r"mod\.info\.cumul == (\d+)", src)
    out["solve_case118"] = {"outer": 20, "cumul": int(cum[1]),
                            "objval": float(re.search(r"objval, ([\d.]+); rtol=1e-6", src).group(1)),
                            "note": "input: tests/golden/case118.m (IEEE 118-bus system written out in MATPOWER layout; the upstream ExaData copy is not available offline)"}
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT)

if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference tree not present")
    main()
