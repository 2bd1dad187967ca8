echo "== shipped"; timeout 120 python scripts/diag_split.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== decision at the top"; EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_waittop.so timeout 120 python scripts/diag_split.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== shipped, window 0"; EXAADMM_HEAVY_WINDOW=0 timeout 120 python scripts/diag_split.py 2>&1 | grep -v amdgpu.ids | tail -12
