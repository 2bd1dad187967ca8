# round 6, session l: the default command after its K / W changed (40 / 5: the region stays on the solve's trajectory), and the rocprofv3 kernel trace of the
# DRIVER's command (--steps 20 --warmup 5) split by phase, so that profiles/ holds the trace of exactly the command BENCH_r06.json comes from
O=gpurun_out/r06_final; mkdir -p $O
R=$PWD; cd /tmp; export TMPDIR=/tmp
PF="--repeats 1 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --no-deep-loop-leg"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_driver_cmd -- python3 $R/bench.py --steps 20 --warmup 5 $PF > $R/$O/stats_bench_driver_cmd.log 2>&1
f=$(find $R/$O/stats_driver_cmd -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $R/scripts/kernel_trace_split.py $f 20 5 300 solve > $R/$O/split_driver_cmd.json && rm -f $f
cd $R
timeout 700 python bench.py > $O/bench_default.log 2>&1; tail -c 400 $O/bench_default.log
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; tail -c 300 $O/bench_driver_cmd.log
cat $O/split_driver_cmd.json | head -60
