cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
timeout 900 python scripts/iter_profile_solve.py case13659pegase 10 > gpurun_out/r04h/iter_profile_solve_13659.txt 2>&1
timeout 900 python scripts/iter_profile_solve.py case_ACTIVSg70k 20 > gpurun_out/r04h/iter_profile_solve_70k.txt 2>&1
timeout 900 python scripts/iter_profile_solve.py case1354pegase 10 > gpurun_out/r04h/iter_profile_solve_1354.txt 2>&1
cat gpurun_out/r04h/iter_profile_solve_13659.txt
tail -5 gpurun_out/r04h/iter_profile_solve_70k.txt
