set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
timeout 2400 python -m pytest tests/test_gpu_dist_threads.py -x -q > gpurun_out/r04d/dist_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04d/dist_tests.log
tail -30 gpurun_out/r04d/dist_tests.log
export EXAADMM_RCCL_LIB=$PWD/tests/rccl_shim/librccl_shim.so RCCL_SHIM_ASYNC=1 GPU_MAX_HW_QUEUES=16
for cfg in "0 1 1" "40 1 1" "100 1 1" "0 0 1" "40 0 1" "100 0 1" "0 1 0" "40 1 0"; do
  set -- $cfg
  RCCL_SHIM_DELAY_US=$1 EXAADMM_DIST_OVERLAP=$2 EXAADMM_DIST_GRAPH=$3 timeout 600 python scripts/dist_overlap_probe.py 2 case13659pegase 204 2>&1 | grep -E "RESULT|Error|error" | tail -2 | sed "s/^/graph=$3 /"
done > gpurun_out/r04d/overlap_probe_async.txt
cat gpurun_out/r04d/overlap_probe_async.txt
