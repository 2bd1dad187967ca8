cd $GRAFT_REPO_ROOT
EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_wide.so timeout 1500 python -m pytest tests/test_gpu_resident.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
run() { # variant case
  if [ $1 = wide ]; then export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_wide.so; else unset EXAADMM_HIP_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 2 --case $2 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 $2 value %.0f whole %.0f kernel_us %.2f' % (r['value'], r['value_reference_metric'], 1e3*r['roofline']['kernel_ms']))"
}
for rep in 1 2 3; do run wide case13659pegase; run base case13659pegase; done
run wide case_ACTIVSg70k; run base case_ACTIVSg70k; run wide case1354pegase; run base case1354pegase
