cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_resident.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
run() { # variant case
  if [ $1 = seq ]; then export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_seq.so; else unset EXAADMM_HIP_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 2 --case $2 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 $2 value %.0f whole %.0f kernel_us %.2f' % (r['value'], r['value_reference_metric'], 1e3*r['roofline']['kernel_ms']))"
}
for rep in 1 2 3; do run new case13659pegase; run seq case13659pegase; done
run new case_ACTIVSg70k; run seq case_ACTIVSg70k; run new case1354pegase; run seq case1354pegase
unset EXAADMM_HIP_LIB
EXAADMM_PROF_RESIDENT=1 EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_prof.so timeout 600 python scripts/resident_timeline.py case13659pegase 300 2>&1 | grep -v amdgpu
