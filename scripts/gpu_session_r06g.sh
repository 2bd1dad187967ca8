# round 6, session g: solver vectors hand-parked in LDS at ONE wavefront per SIMD (p1: 256 VGPR + 52 AGPR instead of + 144, 6 % fewer instructions) against the shipped library
O=gpurun_out/r06g; mkdir -p $O
line() { python3 -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); print('$2: value %.0f it/s  kernel_ms %.4f frac %.3f' % (d['value'], (d['roofline'].get('kernel_ms') or 0), d['roofline'].get('frac') or 0))"; }
for rep in 1 2; do for lib in base p1; do
  [ $lib = base ] && unset EXAADMM_HIP_LIB || export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_$lib.so
  EXAADMM_NO_RESIDENT=1 timeout 300 python bench.py --state prep --steps 200 --warmup 20 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/h_${lib}_$rep.log 2>&1; line $O/h_${lib}_$rep.log "13659 two-launch $lib rep $rep"
  EXAADMM_NO_RESIDENT=1 timeout 300 python bench.py --state prep --case case_ACTIVSg70k --steps 100 --warmup 20 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/k70_${lib}_$rep.log 2>&1; line $O/k70_${lib}_$rep.log "70k two-launch $lib rep $rep"
  timeout 300 python bench.py --periods 4 --steps 100 --warmup 20 --no-cpu-baseline --gpu-seconds 3 > $O/mp4_${lib}_$rep.log 2>&1; line $O/mp4_${lib}_$rep.log "4 periods $lib rep $rep"
done; done | tee $O/ab.txt
for lib in base p1; do
  [ $lib = base ] && unset EXAADMM_HIP_LIB || export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_$lib.so
  timeout 300 python bench.py --case synthetic1M --state prep --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/m1_$lib.log 2>&1; line $O/m1_$lib.log "1M $lib"
done | tee -a $O/ab.txt
EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_p1.so EXAADMM_NO_RESIDENT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "branch_kernel or anchor or trajectory" > $O/tests_p1.txt 2>&1; tail -3 $O/tests_p1.txt
