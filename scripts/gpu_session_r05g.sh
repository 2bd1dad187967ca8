O=gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; echo "suite rc $?" >> $O/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-full-solve --no-north-star > $R/$O/stats_bench.log 2>&1
f=$(find $R/$O/stats -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $R/scripts/kernel_trace_split.py $f 200 20 300 > $R/$O/split.json && rm -f $f
cd $R; tail -3 $O/suite.txt; tail -2 $O/smoke.txt; grep -A8 timed_region $O/split.json
