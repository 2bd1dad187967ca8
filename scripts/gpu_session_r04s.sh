cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
{ timeout 600 python scripts/launch_fixed_cost.py case13659pegase; EXAADMM_NO_RESIDENT=1 timeout 600 python scripts/launch_fixed_cost.py case13659pegase; timeout 600 python scripts/launch_fixed_cost.py case_ACTIVSg70k; EXAADMM_RESIDENT_TILES=0 timeout 600 python scripts/launch_fixed_cost.py case_ACTIVSg70k; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04s/launch_fixed_cost.txt
cat gpurun_out/r04s/launch_fixed_cost.txt
