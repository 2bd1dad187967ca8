#!/usr/bin/env python3
"""Parse the -Rpass-analysis=kernel-resource-usage remarks of the library's own compile (csrc/Makefile writes them next to the
.so) into JSON: registers, spills, LDS, occupancy per kernel, keyed by a readable kernel name, with the digest of the sources.
    python scripts/kernel_resources.py <remarks file> <source digest>  > exaadmm.jl_amd/kernel_resources.json
bench.py copies the dominant kernel's entry into roofline.kernel_resources; profiles/archive_r03_r04/r04_resource_usage.txt is the same table as text."""
import json
import re
import sys

NAMES = [("k_resident_tILi1ELb0ELb0EE", "k_resident"), ("k_resident_tILi1ELb1ELb0EE", "k_resident_t<1,heavy>"), ("k_resident_tILi2ELb0ELb0EE", "k_resident_t<2>"),
         ("k_resident_tILi1ELb0ELb1EE", "k_resident_t<1,peers>"), ("k_resident_tILi1ELb1ELb1EE", "k_resident_t<1,heavy,peers>"), ("k_resident_tILi2ELb0ELb1EE", "k_resident_t<2,peers>"), ("k_p2p_agree", "k_p2p_agree"), ("k_tile_assign", "k_tile_assign"), ("k_tile_keys", "k_tile_keys"), ("k_branch_tILb0ELb1EE", "k_branch_t<false,true>"), ("k_branch_tILb0ELb0EE", "k_branch_t<false,false>"),
         ("k_branch_tILb1ELb0EE", "k_branch_t<true,false>"), ("k_branch_tILb1ELb1EE", "k_branch_t<true,true>"),
         ("k_bus_group_tILb1ELb0EE", "k_bus_group_t<true>"), ("k_bus_group_tILb0ELb0EE", "k_bus_group_t<false>"), ("k_bus_group_tILb1ELb1EE", "k_bus_group_t<true,true>"),
         ("k_gen_ramp", "k_gen_ramp"), ("k_fused_zl", "k_fused_zl"), ("k_halo_unpack", "k_halo_unpack"), ("k_busE", "k_bus")]
FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
          "Occupancy [waves/SIMD]": "waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills_to_agpr",
          "LDS Size [bytes/block]": "lds_bytes_per_block"}


def parse(path):
    out, cur = {}, None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            mangled = m.group(1)
            cur = None
            for key, nice in NAMES:
                if key in mangled:
                    cur = out.setdefault(nice, {})
                    break
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
    return out


if __name__ == "__main__":
    res = parse(sys.argv[1])
    print(json.dumps({"source_digest": sys.argv[2] if len(sys.argv) > 2 else None, "kernels": res}, indent=1, sort_keys=True))
