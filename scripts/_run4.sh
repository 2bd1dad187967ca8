O=gpurun_out/r02_b4
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "native or partitions" > $O/pytest_native.log 2>&1; tail -3 $O/pytest_native.log
python bench.py > $O/bench_default.log 2>&1; tail -c 600 $O/bench_default.log; echo
python bench.py --force-dist --steps 200 --warmup 20 > $O/bench_dist1_native.log 2>&1; tail -c 1500 $O/bench_dist1_native.log; echo
EXAADMM_DIST_PYTHON_LOOP=1 python bench.py --force-dist --steps 200 --warmup 20 > $O/bench_dist1_python.log 2>&1; tail -c 400 $O/bench_dist1_python.log; echo
python bench.py --force-dist --case case_ACTIVSg70k --steps 100 --warmup 20 > $O/bench_dist1_70k.log 2>&1; tail -c 400 $O/bench_dist1_70k.log; echo
rocprofv3 -L > $O/counters_list.txt 2>&1
bash scripts/collect_pmc.sh case13659pegase $O/pmc_case13659pegase 200
bash scripts/collect_pmc.sh synthetic1M $O/pmc_synthetic1M 50
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_solve -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/$O/trace_solve.log 2>&1
cd $R
find $O -name "*.csv" | head -40
