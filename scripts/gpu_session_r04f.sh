cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
export EXAADMM_RCCL_LIB=$PWD/tests/rccl_shim/librccl_shim.so RCCL_SHIM_ASYNC=1 GPU_MAX_HW_QUEUES=16
for cfg in "0 1" "40 1" "0 0" "40 0"; do
  set -- $cfg
  RCCL_SHIM_DELAY_US=$1 EXAADMM_DIST_OVERLAP=$2 timeout 900 python scripts/dist_overlap_probe.py 2 case_ACTIVSg70k 108 2>&1 | grep -E "RESULT|Error|error" | tail -2
done > gpurun_out/r04f/overlap_probe_70k.txt
cat gpurun_out/r04f/overlap_probe_70k.txt
