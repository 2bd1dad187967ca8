set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout 1200 python -m pytest tests/test_gpu_dist_threads.py -x -q -k "hipgraph or one_stream" > gpurun_out/r04c/dist_async_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04c/dist_async_tests.log
tail -30 gpurun_out/r04c/dist_async_tests.log
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q -s -k "mostly_taken" > gpurun_out/r04c/hog.log 2>&1; echo "exit $?" >> gpurun_out/r04c/hog.log
tail -12 gpurun_out/r04c/hog.log
export EXAADMM_RCCL_LIB=$PWD/tests/rccl_shim/librccl_shim.so RCCL_SHIM_ASYNC=1 GPU_MAX_HW_QUEUES=16
for cfg in "0 1" "40 1" "100 1" "0 0" "40 0" "100 0"; do
  set -- $cfg
  RCCL_SHIM_DELAY_US=$1 EXAADMM_DIST_OVERLAP=$2 timeout 600 python scripts/dist_overlap_probe.py 2 case13659pegase 200 2>&1 | grep -E "RESULT|Error|error" | tail -2
done > gpurun_out/r04c/overlap_probe_async.txt
cat gpurun_out/r04c/overlap_probe_async.txt
