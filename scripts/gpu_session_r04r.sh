cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04r
timeout 1500 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py -x -q > gpurun_out/r04r/tests.log 2>&1; echo "exit $?" >> gpurun_out/r04r/tests.log
tail -6 gpurun_out/r04r/tests.log
for c in case13659pegase case_ACTIVSg70k case1354pegase; do
  timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 2 --case $c 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$c value %.0f whole %.0f (%s %s)' % (r['value'], r['value_reference_metric'], r['full_solve']['outer'], r['full_solve']['cumul']))"
done
