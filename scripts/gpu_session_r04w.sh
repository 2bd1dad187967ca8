cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04w
timeout 2400 python -m pytest tests/test_gpu_dist_threads.py -x -q > gpurun_out/r04w/dist_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04w/dist_tests.log
tail -6 gpurun_out/r04w/dist_tests.log
export EXAADMM_RCCL_LIB=$PWD/tests/rccl_shim/librccl_shim.so RCCL_SHIM_ASYNC=1 GPU_MAX_HW_QUEUES=16
for rep in 1 2; do for cfg in "1" "0"; do
  EXAADMM_DIST_FOLD=$cfg timeout 600 python scripts/dist_overlap_probe.py 2 case13659pegase 204 2>&1 | grep -E "RESULT|Error|error" | tail -1 | sed "s/^/fold=$cfg /" | cut -c1-200
done; done > gpurun_out/r04w/fold_probe.txt
for cfg in "1" "0"; do
  EXAADMM_DIST_FOLD=$cfg timeout 900 python scripts/dist_overlap_probe.py 2 case_ACTIVSg70k 108 2>&1 | grep -E "RESULT|Error|error" | tail -1 | sed "s/^/fold=$cfg /" | cut -c1-200
done >> gpurun_out/r04w/fold_probe.txt
cat gpurun_out/r04w/fold_probe.txt
