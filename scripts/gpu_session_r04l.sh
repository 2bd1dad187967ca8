cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
timeout 3400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04l/gpu_suite.log 2>&1; echo "exit $?" >> gpurun_out/r04l/gpu_suite.log
tail -25 gpurun_out/r04l/gpu_suite.log
