O=gpurun_out/r05j; mkdir -p $O
export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_series.so EXAADMM_PROF_RESIDENT=1
timeout 300 python scripts/heavy_series.py case13659pegase 300 40 2>&1 | grep -v amdgpu.ids | tee $O/heavy_series.txt
