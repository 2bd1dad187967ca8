# round 6, final session: the -m gpu suite on the final build, then the final batch (profiles/r06_*), incl. the kernel trace of the DRIVER's command
O=gpurun_out/r06k; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -s > $O/tests.txt 2>&1; tail -6 $O/tests.txt | cut -c1-300
grep -h "branches with other\|outer 2 iteration\|sampled branches\|reference-arithmetic build" $O/tests.txt | tee $O/parity_counts.txt
BUILD_NOTE="round 6 final build (LDS parking, pow outlined, launch cap, 15 options)" bash scripts/final_batch.sh gpurun_out/r06_final r06 2>&1 | tail -20
R=$PWD; cd /tmp; export TMPDIR=/tmp
PF="--repeats 1 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --no-deep-loop-leg"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_final/stats_driver_cmd -- python3 $R/bench.py --steps 20 --warmup 5 $PF > $R/gpurun_out/r06_final/stats_bench_driver_cmd.log 2>&1
f=$(find $R/gpurun_out/r06_final/stats_driver_cmd -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $R/scripts/kernel_trace_split.py $f 20 5 300 solve > $R/gpurun_out/r06_final/split_driver_cmd.json && rm -f $f
cd $R
