O=gpurun_out/r05i; mkdir -p $O
export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_prof.so EXAADMM_PROF_RESIDENT=1
timeout 300 python scripts/heavy_timeline.py case13659pegase 300 2>&1 | grep -v amdgpu.ids | tee $O/heavy_timeline.txt

