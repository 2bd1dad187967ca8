cd $GRAFT_REPO_ROOT
export EXAADMM_PROF_RESIDENT=1
for rep in 1 2; do
EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_merged_prof.so timeout 600 python scripts/resident_timeline.py case13659pegase 300 2>&1 | grep -v amdgpu | sed 's/^/MERGED /'
EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_prof.so timeout 600 python scripts/resident_timeline.py case13659pegase 300 2>&1 | grep -v amdgpu | sed 's/^/SEQ    /'
done
