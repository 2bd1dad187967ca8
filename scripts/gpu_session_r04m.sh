cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
run() { # variant case
  if [ $1 = seq ]; then export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_seq.so; else unset EXAADMM_HIP_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 2 --case $2 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 $2 value %.0f whole %.0f kernel_us %.2f' % (r['value'], r['value_reference_metric'], 1e3*r['roofline']['kernel_ms']))"
}
for rep in 1 2 3; do run merged case13659pegase; run seq case13659pegase; done
run merged case_ACTIVSg70k; run seq case_ACTIVSg70k; run merged case1354pegase; run seq case1354pegase
