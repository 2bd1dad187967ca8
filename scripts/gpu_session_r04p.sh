cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04p
timeout 1500 python -m pytest tests/test_gpu_resident.py -x -q -k "bitwise or twice" > gpurun_out/r04p/resident_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04p/resident_tests.log
tail -4 gpurun_out/r04p/resident_tests.log
run() { # tiles steps
  EXAADMM_RESIDENT_TILES=$1 timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 2 --case case_ACTIVSg70k --steps $2 --warmup 20 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('tiles=$1 steps=$2 value %.0f whole %.0f kernel_us %.2f %s' % (r['value'], r['value_reference_metric'], 1e3*r['roofline']['kernel_ms'], r['roofline']['loop']))"
}
for rep in 1 2; do run 1 100; run 0 100; run 1 200; run 0 200; done
