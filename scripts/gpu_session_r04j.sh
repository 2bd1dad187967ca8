R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04j
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04j/mp -- python3 $R/bench.py --periods 4 --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline --no-full-solve > $R/gpurun_out/r04j/mp_bench.log 2>&1
cd $R
tail -1 gpurun_out/r04j/mp_bench.log | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
f=$(find gpurun_out/r04j/mp -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
find gpurun_out/r04j -name "*kernel_trace.csv" -delete
