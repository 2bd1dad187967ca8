#!/usr/bin/env python3
"""bench.py -- ADMM inner iterations/sec of the MI355X-native ACOPF two-level ADMM inner loop.

    python bench.py --gpus N --steps K --warmup W

One "step" = one inner ADMM iteration (generator + branch TRON + bus + z/l/residual) over the
whole grid resident in HBM.  Workload at N=1: BASELINE.json configs[2], case13659pegase
(rho_pq=5e1 rho_va=5e3 scale=1e-4).  The real MATPOWER file is looked up in $MATPOWER_DATA
and ./data; when absent (no network in this environment) a seeded size-matched synthetic grid
(ngen 4092 / nline 20467 / nbus 13659) is used and the JSON says so.

Benchmark state (round 6): the K timed steps start ON THE TRAJECTORY of the workload's own two-level solve -- by default
at inner iteration 60 of OUTER iteration 2 (`--state solve --state-outer 2 --state-inner 60`): outer iteration 1 runs to its stop test exactly as
admm_two_level does (untimed), lz and beta are updated, 60 iterations of outer iteration 2 follow (untimed: the multipliers' penalties restart at
the top of every inner loop and the first iterations of a loop are the expensive ones -- 0.15 - 0.19 ms each, against 0.07 ms for its last 30; the
stretch behind iteration 60 costs ~0.09 ms, which is what the solve's average iteration costs: 0.10 ms), then the W warm-up and the K timed
steps -- fixed count, no early break.  Rounds 1-5 timed a state the solve never visits on this workload (300 iterations deep into outer iteration 1, whose loop stops
after 96): that figure -- what a solve sees whose inner loops run into the hundreds, like the real pegase inputs' -- is still measured and
reported as `deep_loop_state` (`--state prep --prep 300` makes it the headline again).  The CPU baseline is timed from the same state as `value`.
What a whole solve sees is the `full_solve` block (the reference's own definition of the metric).

For N>1 (launched by torch.distributed.run) the grid is N areas of that size joined by tie
lines, one area per GPU (weak scaling); see exaadmm.jl_amd/distributed.py and DESIGN.md.

`full_solve` (default on) is the reference's own definition of the metric: cumul / time_overall of a whole
solve_acopf with the README settings (outer 20 x inner 1000, early breaks on), with the CPU port timed on the same
solve (whole solve when it fits 30 s, else its first iterations -- said in `cpu_port.sample`).  `value` stays the
fixed-count figure the K / W contract defines.

Prints ONE JSON line (rank 0).  `roofline` refers to the dominant kernel (k_resident, the resident loop's one launch per
batch of iterations, where every wavefront of the grid fits the chip; k_branch of the two-launch loop otherwise):
achieved = algorithmic bytes per launch / mean launch duration from HIP events on the launch
stream.  `cpu_baseline` = the C oracle (serial port of the reference CPU path) timed on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# algorithmic HBM bytes (SURVEY.md §8d, FP64 values, int32 indices)
BYTES_PER_BRANCH_SOLVE = 568.0
BYTES_PER_ITER = lambda G, L, B: 340.0 * G + 1408.0 * L + 44.0 * B

CASE = "case13659pegase"
CASE_PARAMS = {"case13659pegase": dict(rho_pq=5e1, rho_va=5e3, scale=1e-4),
               "case1354pegase": dict(rho_pq=1e1, rho_va=1e3, scale=1e-4),
               "case_ACTIVSg70k": dict(rho_pq=3e4, rho_va=3e5, scale=1e-5),
               "synthetic1M": dict(rho_pq=5e1, rho_va=5e3, scale=1e-4),
               # the reference's own test cases (real input files under tests/golden/), with the settings of its anchors
               # (test/algorithms/acopf_update_cpu.jl:168-179: outer_iterlim=25, outer_eps=2e-5, no bound tightening)
               "case9": dict(rho_pq=4e2, rho_va=4e4, scale=1e-4, tight_factor=1.0, outer_iterlim=25, outer_eps=2e-5),
               "case118": dict(rho_pq=4e2, rho_va=4e4, scale=1e-4, tight_factor=1.0, outer_iterlim=25, outer_eps=2e-5)}


def find_case_file(name):
    for d in (os.environ.get("MATPOWER_DATA", ""), os.path.join(ROOT, "data"), os.path.join(ROOT, "tests", "golden")):
        if d:
            p = os.path.join(d, name + ".m")
            if os.path.exists(p):
                return p
    return None


def cpu_baseline(grid, params, budget_s=15.0, nthreads=1, prep=100, state=None):
    """Time the oracle (kind=port) on a bounded number of inner iterations.  nthreads=1 is the
    reference's own serial structure; nthreads>1 adds OpenMP over branches (extra information).
    state = (outer, inner): the same point of the solve's trajectory as the GPU run (outer iterations 1 .. outer-1 to their stop
    tests, `inner` iterations of the last one), else `prep` iterations of outer iteration 1."""
    import math
    from oracle import oracle as O
    m = O.OracleModel(grid, params["rho_pq"], params["rho_va"], scale=params["scale"], nthreads=nthreads)

    def step(it):
        m.inner_prestep(); m.update_x(it); m.update_xbar(); m.update_z(); m.update_l()
        return m.update_residual()

    it, untimed = 0, 0
    if state is not None:
        par = m.par
        sqrt_d = math.sqrt(m.nvar)
        beta = par.initial_beta
        for outer in range(1, state[0]):
            m.info.outer = outer
            nzp = m.norm_z()
            par.beta = beta
            res, it = None, 0
            while it < 1000:
                it += 1
                untimed += 1
                res = step(it)
                if res[0] <= sqrt_d / (2500.0 * outer):
                    break
            m.update_lz()
            if res[2] > par.theta * nzp:
                beta = min(par.inc_c * beta, 1e24)
        m.info.outer = state[0]
        par.beta = beta
        it = 0
        for _ in range(state[1]):
            it += 1
            untimed += 1
            step(it)
        where = "from inner iteration %d of outer iteration %d of the solve" % (state[1], state[0])
    else:
        m.info.outer = 1
        # same phase of the algorithm as the GPU run: the same untimed preparation iterations first
        for _ in range(max(prep, 2)):
            it += 1
            untimed += 1
            step(it)
        where = "after %d untimed preparation iterations" % untimed
    n_done, t_used = 0, 0.0
    while t_used < budget_s and n_done < 500:
        it += 1
        t0 = time.perf_counter()
        step(it)
        t_used += time.perf_counter() - t0
        n_done += 1
    return {"value": n_done / t_used, "unit": "ADMM inner iterations/s", "cores": nthreads, "kind": "port",
            "sample": "%d inner iterations (%s; %d untimed iterations in front) of the same grid, C oracle with %d thread(s), %.1f s"
                      % (n_done, where, untimed, nthreads, t_used)}


def cpu_full_solve(grid, params, budget_s=30.0, nthreads=1, outer_lim=20, outer_eps=2e-4):
    """The reference's own metric (print_statistics.jl:12: cumul / time_overall of admm_two_level.jl:28-80) for the
    CPU port: the oracle runs THE SAME two-level solve (outer 20 x inner 1000, early breaks on) step by step until it
    finishes or the time budget is spent; a cut-off run is a bounded sample (the first iterations of the solve)."""
    import math
    from oracle import oracle as O
    m = O.OracleModel(grid, params["rho_pq"], params["rho_va"], scale=params["scale"], nthreads=nthreads)
    par = m.par
    par.outer_eps = outer_eps
    sqrt_d = math.sqrt(m.nvar)
    beta, cumul, status, outer, finished = par.initial_beta, 0, "IterationLimit", 0, True
    t0 = time.perf_counter()
    while outer < outer_lim:
        outer += 1
        m.info.outer = outer
        nzp = m.norm_z()
        par.beta = beta
        inner, eps = 0, sqrt_d / (2500.0 * outer)
        while inner < 1000:
            inner += 1
            cumul += 1
            m.inner_prestep(); m.update_x(inner); m.update_xbar(); m.update_z(); m.update_l()
            res = m.update_residual()
            if res[0] <= eps:
                break
            if time.perf_counter() - t0 > budget_s:
                finished = False
                break
        if not finished:
            break
        if res[3] <= sqrt_d * par.outer_eps:
            status = "Solved"
            break
        m.update_lz()
        if res[2] > par.theta * nzp:
            beta = min(par.inc_c * beta, 1e24)
    dt = time.perf_counter() - t0
    return {"iterations_per_s": cumul / dt, "cumul": cumul, "time_s": dt, "cores": nthreads, "kind": "port",
            "finished": finished, "status": status if finished else "cut off by the time budget", "outer": outer,
            "objval": m.objective() if finished else None,
            "sample": ("the whole solve" if finished else "the first %d inner iterations of the same solve (%.0f s budget)" % (cumul, budget_s))}


PMC_FILES = (os.path.join("profiles", "r06_pmc_summary.json"), os.path.join("profiles", "r05_pmc_summary.json"))


def pmc_entry(workload, kernel_key, lib_version):
    """HBM traffic / VALU utilisation of a workload's dominant kernel from the committed rocprofv3 --pmc summary (scripts/collect_pmc.sh +
    scripts/summarize_pmc.py; bench.py itself cannot collect counters).  An entry taken from other code than the library loaded now is refused.
    Returns (entry or {}, traffic_source, the whole summary or {})."""
    pmc_file = next((f for f in PMC_FILES if os.path.exists(os.path.join(ROOT, f))), PMC_FILES[0])
    try:
        pmc = json.load(open(os.path.join(ROOT, pmc_file)))
    except Exception as e:
        return {}, "none: %s unreadable (%s)" % (pmc_file, e), {}
    kb = pmc.get(workload, {}).get(kernel_key, {})
    if kb and pmc.get("library_version") != lib_version:
        return {}, "refused: %s was collected with %r, the loaded library is %r" % (pmc_file, pmc.get("library_version"), lib_version), pmc
    if kb:
        return kb, "%s (library %s, repository commit %s)" % (pmc_file, pmc.get("library_version"), pmc.get("commit")), pmc
    return {}, "none: %s has no entry for this workload" % pmc_file, pmc


def single_gpu_leg(pkg, syn, args, case, steps, gpu_seconds=2.0):
    """One more workload on the same GPU, bounded to a few seconds of GPU time, with the headline line's state definition and timing
    (fixed-count regions of `steps` iterations from the state after --prep + --warmup iterations, every region from that same state,
    median) and its own roofline block (dominant kernel timed by HIP events on the launch stream over one more region):
      case_ACTIVSg70k -- the N = 1 point of north_star's weak-scaling configuration: ONE area of ACTIVSg70k's size (88 207 branches);
                         `bench.py --gpus N` carries the N-area value of the same workload in the same field;
      synthetic1M     -- BASELINE.json configs[4], the throughput / roofline stress (1M branches, 1.3 GB of state: HBM-resident)."""
    import torch
    params = CASE_PARAMS[case]
    t_all = time.perf_counter()
    grid = syn.sized_like(case, seed=0)
    t_grid = time.perf_counter() - t_all
    env = pkg.AdmmEnv(case, params["rho_pq"], params["rho_va"], tight_factor=0.99, gpu_no=0, verbose=0, grid=grid)
    mod = pkg.ModelAcopf(env)
    env.params.scale = params["scale"]
    mod.sync_params()
    beta = env.params.initial_beta
    mod.run_fixed(args.prep + args.warmup, first_major_iter=1, beta=beta) if args.prep + args.warmup > 0 else None
    mod.state_save()
    first = args.prep + args.warmup + 1

    def region(time_branch=False):
        mod.state_restore()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, ms_b = mod.run_fixed(steps, first_major_iter=first, beta=beta, time_branch=time_branch)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ms_b

    regions = [region()[0]]
    for _ in range(max(0, min(2000, int(round(gpu_seconds / max(regions[0], 1e-6)))) - 1)):
        regions.append(region()[0])
    dt = sorted(regions)[len(regions) // 2]
    _, ms_branch = region(time_branch=True)
    form = mod.resident_form()
    resident = form["tiles"] > 0
    G, L, B = int(grid["ngen"]), int(grid["nline"]), int(grid["nbus"])
    its = steps / dt
    workload = "synthetic-%s-sized" % case
    lib_version = pkg.library().exaadmm_version().decode()
    kb, traffic_source, _ = pmc_entry(workload, "k_resident" if resident else "k_branch", lib_version)
    launch_bytes = float(BYTES_PER_ITER(G, L, B)) if resident else 824.0 * L + 192.0 * G
    achieved = launch_bytes / (ms_branch * 1e-3) / 1e9
    out = {"metric": "ADMM inner iterations/sec", "value": its, "unit": "iterations/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * dt / steps, "scaling": "weak", "repeats": len(regions),
           "config": {"workload": workload, "ngen": G, "nline": L, "nbus": B, "rho_pq": params["rho_pq"], "rho_va": params["rho_va"],
                      "scale": params["scale"], "parallelism": "1 GPU", "resident_form": form, "loop_kind": mod.loop_kind(),
                      "state": "ADMM state after %d untimed iterations from init_solution! (outer 1)" % (args.prep + args.warmup)},
           "roofline": {"bound": "hbm",
                        "kernel": ("k_resident_t<%d%s> (one launch = the %d iterations of a region; figures per iteration)" % (form["tiles"], ", heavy" if form["heavy"] else "", steps))
                                  if resident else "k_branch_t<false,true> (branch solves + generator blocks + the z / l / residual pass of the previous iteration)",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": kb.get("traffic_bytes"), "traffic_source": traffic_source, "valu_busy": kb.get("valu_busy"),
                        "algorithmic_bytes_per_launch": launch_bytes, "kernel_ms": ms_branch,
                        "loop": "resident (one launch per region)" if resident else "two launches per iteration",
                        "iteration_bytes": BYTES_PER_ITER(G, L, B), "iteration_frac": BYTES_PER_ITER(G, L, B) * its / 1e9 / HBM_PEAK_GBS},
           "selfchecks": mod.resident_selfchecks(), "loop_downgrades": mod.loop_downgrades(),
           "seconds_for_this_leg": None, "seconds_generating_the_grid_on_the_host": t_grid, "gpu_seconds": sum(regions)}
    mod.close()
    out["seconds_for_this_leg"] = time.perf_counter() - t_all
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: a region that stays ON the trajectory -- outer iteration 2 of the headline stand-in stops after 109 iterations; the state is its
    #  60th, so 5 + 40 more are iterations the solve itself makes.  Rounds 1-5 defaulted to 20 + 200 from a deep-loop state)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--case", default=CASE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-solve", action="store_true",
                    help="skip the whole-solve leg (cumul / time_overall of solve_acopf, outer 20 x inner 1000, early breaks on)")
    ap.add_argument("--cpu-solve-budget", type=float, default=30.0,
                    help="seconds the CPU port may spend on the whole solve of the full_solve leg (cut off = bounded sample)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU code path even with one rank")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = N areas of the case's size, one per GPU (default; --case case_ACTIVSg70k is north_star's "
                         "target configuration); strong = ONE grid of the case's size sharded N ways")
    ap.add_argument("--prep", type=int, default=300,
                    help="untimed preparation iterations that define the benchmark state (the ADMM state after PREP inner "
                         "iterations from init_solution!): the first iterations of a cold start are several times heavier "
                         "than the thousands that follow in a solve; 0 = time from the cold start")
    ap.add_argument("--state", choices=("solve", "prep"), default="solve",
                    help="benchmark state of the headline figure: solve = a point ON the trajectory of the workload's own two-level solve "
                         "(--state-outer / --state-inner; outer iterations before it run to their stop tests, untimed); prep = the state after "
                         "--prep iterations of outer iteration 1 (rounds 1-5; on the stand-in grids a state the solve never visits -- still "
                         "reported as `deep_loop_state`)")
    ap.add_argument("--state-outer", type=int, default=2)
    ap.add_argument("--state-inner", type=int, default=60,
                    help="(60: on the headline stand-in the stretch of outer iteration 2 whose iterations cost what the whole solve's average iteration "
                         "costs -- ~0.09 against 0.10 ms; the 40 iterations before it are the solve's most expensive, the last 30 its cheapest)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="how many times the K-step timed region is repeated (the median is reported); 0 = as many as fill "
                         "--gpu-seconds of GPU time")
    ap.add_argument("--gpu-seconds", type=float, default=6.0,
                    help="GPU time to spend on the repeated timed region (all GPU legs run BEFORE the CPU legs, long enough for the "
                         "driver's utilisation sampling to see them)")
    ap.add_argument("--periods", type=int, default=1,
                    help="T > 1: multi-period ACOPF (SURVEY.md §8 f4) over T load levels of the same grid, one time-expanded device model")
    ap.add_argument("--no-north-star", action="store_true",
                    help="skip the `north_star_scaling` block of the line (areas of ACTIVSg70k's size, north_star's weak-scaling configuration: "
                         "one area on one GPU at N = 1, N areas over N GPUs otherwise; bounded to about a minute)")
    ap.add_argument("--no-deep-loop-leg", action="store_true", help="skip the `deep_loop_state` block (the rounds 1-5 definition of the headline figure)")
    ap.add_argument("--no-throughput-leg", action="store_true",
                    help="skip the `throughput_1M` block of the line (BASELINE.json configs[4]: synthetic 1M branches / 500k buses on ONE GPU, "
                         "bounded to about 2 s of GPU time; generating the grid on the host takes the better part of a minute)")
    ap.add_argument("--attempt-budget", type=float, default=240.0,
                    help="N > 1: seconds a measuring child process may take before its supervisor ends it and starts the next, more "
                         "conservative mode of the distributed loop (exaadmm.jl_amd/bench_supervisor.py)")
    ap.add_argument("--north-star-budget", type=float, default=150.0, help="N > 1: the same for the north_star_scaling leg")
    ap.add_argument("--leg", default="main", help=argparse.SUPPRESS)   # set by the supervisor for its children
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU).  This parent has
        # not touched the GPU (no torch.cuda / HIP call yet), it only forwards the children's output and exit code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
        raise SystemExit(subprocess.call(cmd))
    if world > 1 and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with --nproc-per-node equal to --gpus" % (args.gpus, world))
    if args.gpus > 1 or world > 1 or args.force_dist:
        import exaadmm_loader
        if os.environ.get("EXAADMM_BENCH_CHILD") != "1":
            # This process -- one per rank, started by the launcher -- only SUPERVISES: it never touches the GPU, runs the measurement in a
            # child process, ends a child that hangs or fails together with its peers' and starts fresh children in the next, more
            # conservative mode of the distributed loop (graph -> direct -> one stream -> Python twin -> gloo); rank 0 prints the line of
            # the first attempt that completes on every rank, with `attempts` and the `north_star_scaling` leg (N areas of ACTIVSg70k's size).
            sup = exaadmm_loader.load_submodule("bench_supervisor")
            legs = [{"name": "main", "budget_s": args.attempt_budget}]
            if not args.no_north_star and args.case != "case_ACTIVSg70k" and args.periods == 1:
                legs.append({"name": "north_star_scaling", "budget_s": args.north_star_budget})
            me = os.path.abspath(__file__)

            def make_cmd(leg):
                cmd = [sys.executable, me] + sys.argv[1:]
                if leg["name"] == "north_star_scaling":
                    cmd += ["--case", "case_ACTIVSg70k", "--scaling", "weak", "--leg", "north_star_scaling", "--no-cpu-baseline"]
                return cmd

            line, rc = sup.supervise(make_cmd, rank, world, local_rank, legs, budget_s=args.attempt_budget)
            if rank == 0 and line is not None:
                sys.stdout.flush()
                print(json.dumps(line), flush=True)
            raise SystemExit(rc)
        dist_mod = exaadmm_loader.load_submodule("distributed")
        return dist_mod.bench_main(args, rank, world, local_rank)

    import torch
    import exaadmm_loader
    pkg = exaadmm_loader.load_package()
    syn = exaadmm_loader.load_submodule("synthetic")
    pkg.library()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the product has no CPU path")

    params = CASE_PARAMS[args.case]
    tight = params.get("tight_factor", 0.99)
    outer_lim, outer_eps = params.get("outer_iterlim", 20), params.get("outer_eps", 2e-4)
    path = find_case_file(args.case)
    if path:
        grid = pkg.opf_loaddata(path, tight_factor=tight)
        workload, data = args.case, "real MATPOWER file"
    else:
        grid = syn.sized_like(args.case, seed=0)
        workload, data = "synthetic-%s-sized" % args.case, "synthetic"
    G, L, B = int(grid["ngen"]), int(grid["nline"]), int(grid["nbus"])

    env = pkg.AdmmEnv(args.case, params["rho_pq"], params["rho_va"], tight_factor=tight, gpu_no=0, verbose=0, grid=grid)
    T = max(1, args.periods)
    if T > 1:
        import numpy as np
        lvl = 1.0 + 0.01 * np.sin(np.arange(T) * 2 * np.pi / max(T, 2))   # smooth load profile, +-1 %
        mp_load = (np.stack([env.data["Pd"] * a for a in lvl], 1), np.stack([env.data["Qd"] * a for a in lvl], 1))
        mod = pkg.ModelMpacopf(env, mp_load, end_period=T)
        workload += "-x%d-periods" % T
    else:
        mod = pkg.ModelAcopf(env)
    env.params.scale = params["scale"]
    mod.sync_params()
    beta = env.params.initial_beta

    # benchmark state (module docstring): a point on the trajectory of the workload's own solve (default), or PREP inner iterations of outer
    # iteration 1 from init_solution!; then the W warm-up steps (untimed), then exactly K timed steps
    on_trajectory = (args.state == "solve" and T == 1)
    done = 0
    state_desc = None
    if on_trajectory:
        import math
        par = env.params
        sqrt_d = math.sqrt(mod.nvar)
        loops = []
        for outer in range(1, args.state_outer):   # admm_two_level.jl:29-77, the inner loop fused (exaadmm_run_inner)
            par.beta = beta
            mod.info.outer = outer
            pkg.admm_outer_prestep(env, mod)
            inner, res = mod.run_inner(outer, 1000, beta)
            loops.append(inner)
            if res[3] <= sqrt_d * outer_eps:
                raise SystemExit("bench.py: the solve ends in outer iteration %d (%s inner iterations): --state-outer %d is not on its trajectory"
                                 % (outer, loops, args.state_outer))
            pkg.admm_update_lz(env, mod)
            if res[2] > par.theta * mod.info.norm_z_prev:
                beta = min(par.inc_c * beta, 1e24)
        par.beta = beta
        mod.info.outer = args.state_outer
        if args.state_inner > 0:
            mod.run_fixed(args.state_inner, first_major_iter=1, beta=beta)
            done += args.state_inner
        state_desc = ("inner iteration %d of outer iteration %d of the workload's own two-level solve (outer iteration(s) before it ran %s inner "
                      "iterations to their stop tests; beta %.0f)" % (args.state_inner, args.state_outer, loops, beta))
    elif args.prep > 0:
        mod.run_fixed(args.prep, first_major_iter=1, beta=beta)
        done += args.prep
    if args.warmup > 0:
        mod.run_fixed(args.warmup, first_major_iter=done + 1, beta=beta)
        done += args.warmup
    # timed region: exactly K steps, nothing but the iteration's own kernels on the stream, bracketed by synchronisation on
    # both sides.  (Round 1 bracketed every branch launch of the timed region with a HIP event pair: the rocprofv3 trace
    # shows 5.7 us of idle queue on either side of the kernel for them -- profiles/archive_r01_r02/r02_a_kernel_gaps_bench_with_events.txt.)
    # The region is REPEATED -- every repeat from THE SAME ADMM state (a device-side snapshot taken after the preparation and
    # warm-up iterations is restored, untimed, before each one: the iteration keeps getting cheaper as the inner loop
    # converges, 13.1 k it/s at iteration 320 and 16.3 k at 13 000, so consecutive stretches would not measure one thing)
    # -- until the GPU has been busy for about a second, and the MEDIAN region is reported: K = 20 steps last 1.6 ms, too
    # short for the driver's smi sampling to see and at the mercy of one scheduling hiccup.
    # The saved state includes the per-branch solver-trip counters and the trip history from which a resident launch chooses its
    # heavy wavefronts / pairs its tiles (exaadmm_state_save, round 5): without them a repeat chose them from what the PREVIOUS
    # repeat had left behind -- the stragglers of its own future -- and ran 6.6 - 14 % faster than the first pass (DESIGN.md section 4,
    # profiles/r05_honest_replays.txt).  Every repeat is now the first pass.
    can_snapshot = True   # (multi-period handles snapshot their ramp blocks as well)
    if can_snapshot:
        mod.state_save()

    def region(first):
        if can_snapshot:
            mod.state_restore()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nrm, ms, _ = mod.run_fixed(args.steps, first_major_iter=first, beta=beta, time_branch=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, nrm, ms

    regions = []
    dt, norms, ms_total = region(done + 1)
    regions.append(dt)
    repeats = 1 if args.repeats == 1 else (args.repeats if args.repeats > 1 else max(1, min(20000, int(round(args.gpu_seconds / max(dt, 1e-6))))))
    for _ in range(repeats - 1):
        if not can_snapshot:
            done += args.steps
        dt_i, norms, ms_total = region(done + 1)
        regions.append(dt_i)
    done += args.steps
    dt = sorted(regions)[len(regions) // 2]
    if can_snapshot:
        mod.state_restore()   # the kernel-duration leg below starts from the same state as well
    # the dominant kernel's duration: HIP events around each of its launches on the launch stream, over the K steps
    # that FOLLOW the timed region (same steady state), so that the events do not sit inside the figure they explain
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    _, _, ms_branch = mod.run_fixed(args.steps, first_major_iter=done + 1, beta=beta, time_branch=True)
    torch.cuda.synchronize()
    dt_with_events = time.perf_counter() - t1
    # what that run launched, as the library decided it (exaadmm_resident_form): tiles per resident wavefront, heavy wavefronts
    form = mod.resident_form() if T == 1 else {"tiles": 0, "heavy": 0, "launches": 0, "wavefronts": 0}
    # the largest number of solver-loop trips any branch made in the last x-update of that run: the launch is its slowest branch
    # (roofline.critical_path below)
    max_trips = int(mod.branch_counters()[2].max()) if T == 1 else None

    its = args.steps / dt
    if T > 1:
        a_it = m_it = None   # branch counters of the time-expanded model are not broken down per period
        G, L, B = G * T, L * T, B * T   # algorithmic bytes of the time-expanded model (ramp block: +88 B per generator and period)
    else:
        a_it, m_it = mod.branch_stats()
    # HBM traffic of the dominant kernel: rocprofv3 --pmc passes of THIS command (scripts/collect_pmc.sh; bench.py itself
    # cannot collect counters), summarised per workload in profiles/archive_r03_r04/r03_pmc_summary.json together with the version string
    # of the library they were taken from.  An entry taken from other code than the library loaded now is refused.
    # which loop ran: 2 = the resident loop (ONE launch for the K steps of a region: branch solves, bus updates by the last lane
    # to arrive, residual sums and stop test inside; the figures below are per ITERATION of that launch), 1 = two launches per iteration
    resident = (T == 1 and form["tiles"] > 0)
    lib_version = pkg.library().exaadmm_version().decode()
    kb, traffic_source, pmc = pmc_entry(workload, "k_resident" if resident else "k_branch", lib_version)
    traffic, valu_busy, insts_per_trip = kb.get("traffic_bytes"), kb.get("valu_busy"), kb.get("valu_insts_per_wave_trip")
    pmc_file = traffic_source.split(" ")[0] if not traffic_source.startswith(("none", "refused")) else PMC_FILES[0]
    # (the instruction count of a solver trip belongs to the CODE: where this workload has no calibration pass of its own, the headline
    #  workload's -- taken from the same library -- stands in, and `instructions_calibrated_on` says so; `traffic_source` only ever names
    #  where `traffic` came from)
    insts_calibrated_on = workload
    if insts_per_trip is None and pmc.get("library_version") == lib_version:
        insts_per_trip = pmc.get("synthetic-case13659pegase-sized", {}).get("k_resident", {}).get("valu_insts_per_wave_trip")
        insts_calibrated_on = "synthetic-case13659pegase-sized"
    occ = mod.branch_kernel_occupancy() if hasattr(mod, "branch_kernel_occupancy") else None
    # registers / spills / LDS of the dominant kernel, from the remarks of the library's own compile (csrc/Makefile ->
    # exaadmm.jl_amd/kernel_resources.json, which carries the digest of the sources it belongs to)
    # which form of the resident kernel ran: asked from the library (exaadmm_resident_form), not re-derived here
    two_tile = form["tiles"] == 2
    heavy = form["heavy"] > 0
    kernel_resources = None
    try:
        kr = json.load(open(os.path.join(ROOT, "exaadmm.jl_amd", "kernel_resources.json")))
        if kr.get("source_digest") and ("src:" + kr["source_digest"]) in lib_version:
            kernel_resources = kr["kernels"].get(("k_resident_t<2>" if two_tile else ("k_resident_t<1,heavy>" if heavy else "k_resident")) if resident else (("k_branch_t<true,true>" if mod.loop_kind() == 1 else "k_branch_t<true,false>") if T > 1 else "k_branch_t<false,true>"))
    except Exception:
        pass
    branch_bytes = BYTES_PER_BRANCH_SOLVE * L
    fused_zl = mod.loop_kind() >= 1
    if fused_zl:
        # the launch also carries the generator update (128 B per generator) and the z / l / residual pass of the previous
        # iteration; of the latter only what the branch loads do not already bring counts: lz read + z, z_prev, l writes
        # = 32 B per vector element (64 B minimal + 8 B z_prev, minus the 40 B of shared reads)
        branch_bytes = 824.0 * L + 192.0 * G
    if resident:
        branch_bytes = float(BYTES_PER_ITER(G, L, B))   # the launch does whole iterations
    achieved = branch_bytes / (ms_branch * 1e-3) / 1e9
    # The roofline that BINDS at the named sizes.  The launch is its slowest branch: a chain of `max_trips` solver
    # trips, each a dependent sequence of wave64 instructions -- a SIMD issues one FP64 instruction of a wavefront per 4 clocks whatever
    # the number of active lanes -- so per iteration the launch cannot take less than
    #     max_trips x (VALU instructions of one wavefront per solver trip) x 4 clocks / shader clock.
    # instructions per trip: SQ_INSTS_VALU of the kernel / the wavefronts' trips, from the rocprofv3 --pmc pass summarised in
    # profiles/r05_pmc_summary.json (scripts/summarize_pmc.py); shader clock: the device's (hipDeviceProp clockRate).
    critical = None
    if T == 1 and max_trips:
        try:
            clk_khz = float(torch.cuda.get_device_properties(0).clock_rate)
        except Exception:
            clk_khz = 2.4e6
        if insts_per_trip:
            floor_ms = max_trips * insts_per_trip * 4.0 / (clk_khz * 1e3) * 1e3
            critical = {"bound": "fp64-issue chain of the slowest branch", "max_solver_trips": max_trips, "valu_insts_per_wave_trip": insts_per_trip,
                        "instructions_calibrated_on": insts_calibrated_on, "clocks_per_instruction": 4, "shader_clock_mhz": clk_khz / 1e3, "floor_ms": floor_ms, "kernel_ms": ms_branch,
                        "frac": floor_ms / ms_branch, "definition": "max_solver_trips x valu_insts_per_wave_trip x 4 clocks / shader clock, over kernel_ms (per iteration)"}
        else:
            critical = {"max_solver_trips": max_trips, "shader_clock_mhz": clk_khz / 1e3, "frac": None,
                        "note": "no valu_insts_per_wave_trip for this workload / library in " + pmc_file}
    out = {
        "metric": "ADMM inner iterations/sec", "value": its, "unit": "iterations/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": data,
        "value_definition": "K / (median wall time of a K-step timed region): fixed-count inner iterations, "
                            "no early break, from the ADMM state config.state names (+ the W warm-up steps); the region was timed `repeats` "
                            "times, every time from that same state (device-side snapshot restored, untimed).  The reference's own metric (cumul / time_overall of a whole solve_acopf, "
                            "print_statistics.jl:12) is value_reference_metric / full_solve",
        "repeats": len(regions),
        "regions": {"first": args.steps / regions[0], "median": its, "last": args.steps / regions[-1],
                    "min": args.steps / max(regions), "max": args.steps / min(regions), "unit": "iterations/s",
                    "gpu_seconds": sum(regions)},
        "config": {"workload": workload, "ngen": G, "nline": L, "nbus": B, "rho_pq": params["rho_pq"],
                   "rho_va": params["rho_va"], "scale": params["scale"], "tight_factor": tight,
                   "mode": "fixed-count inner iterations, no early break", "parallelism": "1 GPU",
                   "state": state_desc if on_trajectory else "ADMM state after %d untimed preparation iterations from init_solution! (outer 1, beta %.0f)" % (args.prep, beta)},
        "roofline": {"bound": "hbm", "kernel": (("k_resident_t<2> (up to two tiles of 64 branches per resident wavefront; " if two_tile else ("k_resident_t<1, heavy> (the branches with the most solver trips on wavefronts of their own; " if heavy else "k_resident (")) + "one launch = the K iterations of a region; figures per iteration)" if resident else
                                                "k_branch" + (" (with the fused z/l prologue)" if fused_zl else "")), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": branch_bytes, "kernel_ms": ms_branch,
                     "kernel_ms_measured_over": ("HIP events around the k_resident launch of the %d steps following the timed region, divided by %d" % (args.steps, args.steps)) if resident
                                                else "HIP events around each k_branch launch of the %d steps following the timed region" % args.steps,
                     "loop": "resident (one launch per inner loop / per fixed-count run)" if resident else "two launches per iteration",
                     "iteration_bytes": BYTES_PER_ITER(G, L, B),
                     "iteration_frac": BYTES_PER_ITER(G, L, B) * its / 1e9 / HBM_PEAK_GBS,
                     # HBM is the wrong roof at the named sizes (state is Infinity-Cache resident, the kernel is the FP64
                     # dependent-issue chain of its slowest lane): occupancy and VALU utilisation say how it is bound
                     "waves_per_simd": occ, "waves_launched_per_simd": ((L + 63) // 64) / 1024.0,
                     "kernel_resources": kernel_resources, "resident_form": form, "critical_path": critical,
                     "valu_busy": valu_busy, "traffic_source": traffic_source, "library_version": lib_version,
                     "note": "latency/FP64-issue bound: state is Infinity-Cache resident; traffic / valu_busy come from the "
                             "rocprofv3 --pmc passes of this command (traffic_source); see DESIGN.md"},
        "solver": {"avg_auglag_it": a_it, "avg_tron_minor_it": m_it, "primres": float(norms[0]),
                   "device_ms_total": ms_total, "ms_per_step_with_kernel_events": 1e3 * dt_with_events / args.steps},
    }
    out["solver"]["resident_selfchecks"] = list(mod.resident_selfchecks()) if T == 1 else None
    out["solver"]["loop_downgrades"] = mod.loop_downgrades() if T == 1 else None
    deep_dt = dt
    mod.close()   # (the legs below create their own handles: give the headline handle's HBM back first)
    if on_trajectory and not args.no_deep_loop_leg:
        # the figure rounds 1-5 reported as `value`: the state 300 iterations deep into outer iteration 1 -- what the iterations of a long inner
        # loop cost (real pegase inputs run hundreds of iterations per loop; this stand-in's loops stop after ~100).  A fresh handle.
        try:
            m2 = pkg.ModelAcopf(env)
            env.params.scale = params["scale"]
            m2.sync_params()
            b0 = env.params.initial_beta
            m2.run_fixed(args.prep + args.warmup, first_major_iter=1, beta=b0)
            m2.state_save()
            regs = []
            for _ in range(max(1, min(2000, int(round(2.0 / max(deep_dt, 1e-6)))))):
                m2.state_restore()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m2.run_fixed(args.steps, first_major_iter=args.prep + args.warmup + 1, beta=b0)
                torch.cuda.synchronize()
                regs.append(time.perf_counter() - t0)
            d2 = sorted(regs)[len(regs) // 2]
            out["deep_loop_state"] = {"value": args.steps / d2, "unit": "iterations/s", "ms_per_step": 1e3 * d2 / args.steps, "repeats": len(regs),
                                      "state": "ADMM state after %d untimed iterations of outer iteration 1 from init_solution! (beta %.0f): the definition of `value` in rounds 1-5" % (args.prep + args.warmup, b0),
                                      "resident_form": m2.resident_form(),
                                      "iteration_frac": BYTES_PER_ITER(G, L, B) * (args.steps / d2) / 1e9 / HBM_PEAK_GBS}
            m2.close()
        except Exception as e:
            out["deep_loop_state"] = {"error": repr(e)}
    if T == 1 and not args.no_north_star and args.case != "case_ACTIVSg70k":
        try:
            out["north_star_scaling"] = single_gpu_leg(pkg, syn, args, "case_ACTIVSg70k", args.steps)
        except Exception as e:   # never lose the line over an extra
            out["north_star_scaling"] = {"error": repr(e)}
    if T == 1 and not args.no_throughput_leg and args.case != "synthetic1M":
        # BASELINE.json configs[4]: the 1M-branch throughput / roofline stress, the only configuration whose state (1.3 GB) is HBM-resident --
        # the regime in which the hbm roofline above is the roof that binds
        try:
            out["throughput_1M"] = single_gpu_leg(pkg, syn, args, "synthetic1M", min(args.steps, 20), gpu_seconds=2.0)
        except Exception as e:
            out["throughput_1M"] = {"error": repr(e)}
    if not args.no_full_solve and T == 1:
        # the reference's own definition of the metric (print_statistics.jl:12): cumul / time_overall of a whole
        # solve_acopf with the README settings (outer 20, inner 1000, early breaks active; timer scope
        # admm_two_level.jl:28-80).  Two solves: the first one also pays graph capture and first-touch costs.
        fs = None
        for rep in range(2):
            tw0 = time.perf_counter()
            env2, mod2 = pkg.solve_acopf(args.case, grid=grid, rho_pq=params["rho_pq"], rho_va=params["rho_va"],
                                         scale=params["scale"], tight_factor=tight, outer_iterlim=outer_lim, outer_eps=outer_eps,
                                         inner_iterlim=1000, verbose=0)
            i2 = mod2.info
            host_u, host_v = mod2.solution.u_curr, mod2.solution.v_curr   # results back in host memory
            tw1 = time.perf_counter()
            cur = {"status": i2.status, "outer": i2.outer, "cumul": i2.cumul, "time_overall_s": i2.time_overall,
                   "iterations_per_s": i2.cumul / i2.time_overall, "ms_per_iteration": 1e3 * i2.time_overall / max(i2.cumul, 1),
                   "objval": i2.objval, "mismatch": i2.mismatch, "avg_auglag_it_last": i2.avg_auglag_it,
                   # boundary-inclusive: host grid arrays in -> handle creation (PCIe upload, CSR / slot tables), initial
                   # point, solve, u and v back in host memory (PCIe download).  Reported, never `value`.
                   "host_to_host_s": tw1 - tw0, "iterations_per_s_host_to_host": i2.cumul / (tw1 - tw0)}
            if fs is None:
                first = cur
            fs = cur
            mod2.close()
        out["value_reference_metric"] = fs["iterations_per_s"]
        fs["first_run_time_overall_s"] = first["time_overall_s"]
        fs["definition"] = "cumul / time_overall of solve_acopf(outer_iterlim=%d, outer_eps=%g, inner_iterlim=1000), second of two solves" % (outer_lim, outer_eps)
        out["full_solve"] = fs
        if not args.no_cpu_baseline:
            cf = cpu_full_solve(grid, params, budget_s=args.cpu_solve_budget, outer_lim=outer_lim, outer_eps=outer_eps)
            fs["cpu_port"] = cf
            fs["speedup_vs_cpu_port"] = fs["iterations_per_s"] / cf["iterations_per_s"]
            if cf["finished"]:   # same solve on both sides: the counts and the objective must agree
                fs["cpu_port_same_iteration_counts"] = bool(cf["cumul"] == fs["cumul"] and cf["outer"] == fs["outer"])
                fs["objective_rel_diff_vs_cpu_port"] = abs(cf["objval"] - fs["objval"]) / max(abs(cf["objval"]), 1e-300)
    if T > 1:
        out["config"]["periods"] = T
        out["config"]["mode"] = "multi-period ACOPF (ModelMpacopf), fixed-count inner iterations"
    if not args.no_cpu_baseline and T > 1:
        from oracle import oracle as O
        om = O.OracleMpModel(grid, mp_load[0], mp_load[1], params["rho_pq"], params["rho_va"], scale=params["scale"])
        om.par.beta = beta
        n_done, t_used, it = 0, 0.0, 0
        skip = max(args.prep, 2)
        while t_used < 15.0 and n_done < 200:
            it += 1
            t1 = time.perf_counter()
            om.inner_prestep(); om.update_x(it); om.update_xbar(); om.update_z(); om.update_l(); om.update_residual()
            if it > skip:
                t_used += time.perf_counter() - t1
                n_done += 1
        out["cpu_baseline"] = {"value": n_done / t_used, "unit": "ADMM inner iterations/s", "cores": 1, "kind": "port",
                               "sample": "%d inner iterations (after %d untimed preparation iterations) of the same %d-period model, serial C oracle, %.1f s" % (n_done, skip, T, t_used)}
        out["speedup_vs_cpu_baseline"] = its / out["cpu_baseline"]["value"]
    elif not args.no_cpu_baseline:
        cb = cpu_baseline(grid, params, prep=args.prep, state=(args.state_outer, args.state_inner + args.warmup) if on_trajectory else None)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = its / cb["value"]
        ncpu = min(os.cpu_count() or 1, 16)
        if ncpu > 1:   # extra: the same port with OpenMP over the branch loop (capped: more threads are slower)
            out["cpu_baseline_all_cores"] = cpu_baseline(grid, params, budget_s=8.0, nthreads=ncpu, prep=args.prep,
                                                         state=(args.state_outer, args.state_inner + args.warmup) if on_trajectory else None)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
