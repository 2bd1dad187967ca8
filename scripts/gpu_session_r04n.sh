cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04n
export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_prof.so EXAADMM_PROF_RESIDENT=1
timeout 600 python scripts/resident_timeline.py case13659pegase 300 > gpurun_out/r04n/timeline_13659.txt 2>&1
timeout 600 python scripts/resident_timeline.py case1354pegase 300 > gpurun_out/r04n/timeline_1354.txt 2>&1
cat gpurun_out/r04n/timeline_13659.txt gpurun_out/r04n/timeline_1354.txt
