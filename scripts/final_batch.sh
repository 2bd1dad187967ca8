#!/bin/bash
# GPU box: the measurements profiles/ is refreshed from.  Order matters: kernel traces and PMC passes first, then
# scripts/refresh_profiles.py writes profiles/<round>_pmc_summary.json (with the version string of THIS build), then the bench lines --
# which read `traffic` / `valu_busy` from that summary and refuse it for any other build.  usage: bash scripts/final_batch.sh OUTDIR [ROUND]
O=${1:-gpurun_out/final}; mkdir -p $O
RND=${2:-r06}
T="timeout 700"
R=$PWD; cd /tmp; export TMPDIR=/tmp
# profiling runs time ONE region (--repeats 1) of the default state (a point on the solve's trajectory) and skip every other leg
PF="--repeats 1 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --no-deep-loop-leg"
for c in "" case1354pegase case_ACTIVSg70k synthetic1M; do
  tag=${c:+_$c}; K=200; [ "$c" = "case_ACTIVSg70k" ] && K=100; [ "$c" = "synthetic1M" ] && K=50
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats$tag -- python3 $R/bench.py ${c:+--case $c} --steps $K --warmup 20 $PF > $R/$O/stats_bench$tag.log 2>&1
  f=$(find $R/$O/stats$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python3 $R/scripts/kernel_trace_split.py $f $K 20 300 solve > $R/$O/split$tag.json && rm -f $f
done
cd $R
bash scripts/collect_pmc.sh case13659pegase $O/pmc_case13659pegase 200 > $O/pmc1.log 2>&1
bash scripts/collect_pmc.sh case1354pegase $O/pmc_case1354pegase 200 > $O/pmc2.log 2>&1
bash scripts/collect_pmc.sh case_ACTIVSg70k $O/pmc_case_ACTIVSg70k 100 > $O/pmc3.log 2>&1
bash scripts/collect_pmc.sh synthetic1M $O/pmc_synthetic1M 50 > $O/pmc4.log 2>&1
# calibration of roofline.critical_path: VALU instructions per wavefront trip = SQ_INSTS_VALU per iteration of the resident launch WITHOUT heavy
# wavefronts / the trips its wavefronts ran (scripts/wave_trips.py at the same state, same switch): the deep-loop state (--state prep), whose
# launches are uniform
mkdir -p $O/pmc_noheavy; ( cd /tmp; EXAADMM_NO_HEAVY=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $R/$O/pmc_noheavy/SQ_INSTS_VALU -- python3 $R/bench.py --state prep --steps 200 --warmup 20 $PF > $R/$O/pmc_noheavy.log 2>&1 )
EXAADMM_NO_HEAVY=1 python scripts/wave_trips.py case13659pegase 320 > $O/wave_trips_noheavy.json 2> $O/wave_trips_noheavy.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +8M -delete
python scripts/refresh_profiles.py $O $RND "${BUILD_NOTE:-round 6 final build}" > $O/refresh1.log 2>&1
cp profiles/${RND}_pmc_summary.json $O/${RND}_pmc_summary.json
$T python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1
$T python bench.py > $O/bench_default.log 2>&1
$T python bench.py --case case118 --no-north-star --no-throughput-leg > $O/bench_case118.log 2>&1
$T python bench.py --case case9 --no-north-star --no-throughput-leg > $O/bench_case9.log 2>&1
$T python bench.py --case case1354pegase --steps 100 --warmup 20 --no-north-star --no-throughput-leg > $O/bench_case1354pegase.log 2>&1
$T python bench.py --case case_ACTIVSg70k --steps 100 --warmup 20 --cpu-solve-budget 200 --no-throughput-leg > $O/bench_case_ACTIVSg70k.log 2>&1
$T python bench.py --case synthetic1M --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_synthetic1M.log 2>&1
$T python bench.py --force-dist --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_dist1.log 2>&1
$T python bench.py --force-dist --case case_ACTIVSg70k --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_dist1_70k.log 2>&1
$T python bench.py --periods 4 --steps 100 --warmup 20 > $O/bench_mp4.log 2>&1
EXAADMM_RESIDENT_TILES=0 $T python bench.py --case case_ACTIVSg70k --steps 100 --warmup 20 --no-cpu-baseline --no-throughput-leg > $O/bench_case_ACTIVSg70k_two_launch.log 2>&1
EXAADMM_NO_RESIDENT=1 $T python bench.py --no-cpu-baseline --no-north-star --no-throughput-leg > $O/bench_two_launch_loop.log 2>&1
$T python bench.py --state prep --no-cpu-baseline --no-north-star --no-throughput-leg > $O/bench_deep_loop_state.log 2>&1
for f in $O/bench_*.log; do echo "$f: $(python -c "
import json
for l in open('$f'):
    if l.startswith('{'):
        d=json.loads(l); fs=d.get('full_solve',{}); print('%.1f it/s  ms/step %.4f frac %.3f traffic %s | deep %s | full solve %s' % (d['value'], d['ms_per_step'], d['roofline'].get('frac') or 0, d['roofline'].get('traffic'), (d.get('deep_loop_state') or {}).get('value'), (fs.get('status'), fs.get('outer'), fs.get('cumul'), fs.get('iterations_per_s'), fs.get('speedup_vs_cpu_port'))))
")"; done
