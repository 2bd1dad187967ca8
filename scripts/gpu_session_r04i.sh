cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_prof.so
echo "== k_branch_t<false,true> (two-launch loop), iteration 300" > gpurun_out/r04i/phase_cycles.txt
timeout 600 python scripts/phase_cycles.py 300 case13659pegase >> gpurun_out/r04i/phase_cycles.txt 2>&1
echo "== k_resident (EXAADMM_PROF_RESIDENT=1), iteration 300" >> gpurun_out/r04i/phase_cycles.txt
EXAADMM_PROF_RESIDENT=1 timeout 600 python scripts/phase_cycles.py 300 case13659pegase >> gpurun_out/r04i/phase_cycles.txt 2>&1
cat gpurun_out/r04i/phase_cycles.txt
