cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
timeout 1500 python -m pytest tests/test_gpu_mpacopf.py -x -q > gpurun_out/r04k/mp_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04k/mp_tests.log
tail -25 gpurun_out/r04k/mp_tests.log
for rep in 1 2; do
 timeout 600 python bench.py --periods 4 --steps 100 --warmup 20 --no-cpu-baseline --gpu-seconds 1 2>&1 | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('fused', r['value'], r['ms_per_step'], r['roofline']['kernel_ms'])"
 EXAADMM_MP_NO_SPEC=1 timeout 600 python bench.py --periods 4 --steps 100 --warmup 20 --no-cpu-baseline --gpu-seconds 1 2>&1 | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('four-launch', r['value'], r['ms_per_step'], r['roofline']['kernel_ms'])"
done
