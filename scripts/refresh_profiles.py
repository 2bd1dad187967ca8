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
         "bench_case_ACTIVSg70k": "bench_case_ACTIVSg70k", "bench_synthetic1M": "bench_synthetic1M", "bench_dist1": "bench_force_dist_one_rank", "bench_dist1_70k": "bench_force_dist_one_rank_case_ACTIVSg70k",
         "bench_mp4": "bench_mp_4_periods", "bench_mp4_four_launch": "bench_mp_4_periods_four_launch_loop_EXAADMM_MP_NO_SPEC",
         "bench_case_ACTIVSg70k_two_launch": "bench_case_ACTIVSg70k_two_launch_loop_EXAADMM_RESIDENT_TILES_0", "bench_two_launch_loop": "bench_two_launch_loop_EXAADMM_NO_RESIDENT", "bench_deep_loop_state": "bench_deep_loop_state_state_prep", "bench_case118": "bench_case118_real_file", "bench_case9": "bench_case9_real_file"}
for k, v in names.items():
    f = os.path.join(src, k + ".log")
    if os.path.exists(f):
        json.dump(line(f), open(os.path.join(ROOT, "profiles", "%s_%s.json" % (rnd, v)), "w"), indent=1)
for case, tag in (("", ""), ("case1354pegase", "_case1354pegase"), ("case_ACTIVSg70k", "_case_ACTIVSg70k"), ("synthetic1M", "_synthetic1M")):
    st = glob.glob(os.path.join(src, "stats" + tag, "*", "*kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(ROOT, "profiles", "%s_kernel_stats%s.csv" % (rnd, tag)))
    sp = os.path.join(src, "split%s.json" % tag)
    if os.path.exists(sp):
        shutil.copy(sp, os.path.join(ROOT, "profiles", "%s_kernel_trace_split%s.json" % (rnd, tag)))
out_p = os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % rnd)
try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
except Exception:
    commit = "(taken on the GPU box from a snapshot without .git; see `git log` of this file)"
import ctypes
_lib = ctypes.CDLL(os.path.join(ROOT, "exaadmm.jl_amd", "libexaadmm_hip.so"))
_lib.exaadmm_version.restype = ctypes.c_char_p
library_version = _lib.exaadmm_version().decode()      # (the build the passes were taken from: the batch runs from this tree)
out = {"_note": "rocprofv3 --kernel-trace --pmc <group> passes (scripts/collect_pmc.sh, one counter group per run), per-launch averages over the timed "
                "region of bench.py --repeats 1 (scripts/summarize_pmc.py).  FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE counts requests in units of 64 B "
                "(profiles/archive_r03_r04/r03_fetch_size_counter_check.txt): x 2 for the plane-contiguous readers, both bounds for the bus kernel.  SQ_* cycle counters in "
                "quad-cycles summed over waves; shader clock = GRBM_GUI_ACTIVE / 8 / kernel time.",
       "build": build, "library_version": library_version,
       "commit": commit}
SIZES = {"case13659pegase": (4092, 20467), "case1354pegase": (260, 1991), "case_ACTIVSg70k": (10390, 88207), "synthetic1M": (50000, 1000000)}
for case, steps in (("case13659pegase", 200), ("case1354pegase", 200), ("case_ACTIVSg70k", 100), ("synthetic1M", 50)):
    d, wl = "pmc_" + case, "synthetic-%s-sized" % case
    G, L = SIZES[case]
    p = os.path.join(src, d)
    if not os.path.isdir(p):
        continue
    out.update(json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "summarize_pmc.py"), p, wl, str(steps)])))
    if "k_resident" in out[wl]:   # resident loop: one launch does whole iterations; figures per iteration
        out[wl]["k_resident"]["algorithmic_bytes"] = None
        out[wl]["k_resident"]["algorithmic_bytes_note"] = ("per iteration of the resident launch: the whole iteration's algorithmic bytes (bench.py BYTES_PER_ITER, SURVEY.md 8d); "
                                                           "on top of them the sum terms / bus results / flags the wavefronts hand to each other (sc1 accesses)")
    if "k_branch" in out[wl]:
        out[wl]["k_branch"]["algorithmic_bytes"] = int(824.0 * L + 192.0 * G)
        out[wl]["k_branch"]["algorithmic_bytes_note"] = ("fused launch: 568 B per branch + 128 B per generator + the unshared part of the z/l pass (32 B per vector element) "
                                                         "= 824 L + 192 G.  On top of that the launch writes the bus records it pushes (128 B per branch, 48 B per generator) "
                                                         "and reads their slot indices (8 B / 4 B): design traffic, not algorithmic -- it replaces 256 B of gathers per branch in the bus launch")
    if "k_bus_group" in out[wl]:
        out[wl]["k_bus_group"]["algorithmic_bytes"] = None
# roofline.critical_path: VALU instructions one wavefront issues per solver trip = SQ_INSTS_VALU per iteration of the resident launch WITHOUT heavy
# wavefronts / the wavefront trips of an iteration at that state (scripts/wave_trips.py).  The figure belongs to the CODE (the solver loop's
# instruction count on the paths the lanes of that grid take), so the headline workload's entry carries it for the forms with heavy wavefronts too.
try:
    wt = line(os.path.join(src, "wave_trips_noheavy.json"))
    sub = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "summarize_pmc.py"), os.path.join(src, "pmc_noheavy"), "x", "200"]))["x"]
    insts = sub["k_resident"]["SQ_INSTS_VALU"]
    wl = "synthetic-case13659pegase-sized"
    cal = {"valu_insts_per_iteration_no_heavy": insts, "wave_trips_per_iteration_no_heavy": wt["wave_trips_per_iteration"],
           "valu_insts_per_wave_trip": insts / wt["wave_trips_per_iteration"], "max_trips_at_that_state": wt["max_trips"],
           "how": "SQ_INSTS_VALU of k_resident (EXAADMM_NO_HEAVY=1, per iteration) / sum over the tiles of their largest lane trip count (scripts/wave_trips.py)"}
    out.setdefault(wl, {}).setdefault("k_resident", {}).update({"valu_insts_per_wave_trip": cal["valu_insts_per_wave_trip"], "critical_path_calibration": cal})
except Exception as e:
    out["critical_path_calibration_error"] = repr(e)
json.dump(out, open(out_p, "w"), indent=1)
print("profiles refreshed from", src)
