# round 6, session j: solver vectors hand-parked in LDS in EVERY kernel that carries the lane code (libexaadmm_hip_lp.so, -DEXA_LDS_PARK: k_resident 150 -> 58 AGPRs,
# heavy form 170 -> 83) against the shipped library: resident / parity tests under the variant, interleaved A/B of the headline
O=gpurun_out/r06j; mkdir -p $O
LP=$PWD/exaadmm.jl_amd/libexaadmm_hip_lp.so
EXAADMM_HIP_LIB=$LP timeout 1500 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py -m gpu -x -q -k "not hooks and not test_build and not two_processes" > $O/tests_lp.txt 2>&1; tail -4 $O/tests_lp.txt | cut -c1-300
line() { python3 -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); fs=d.get('full_solve') or {}; print('$2: value %.0f it/s  kernel_ms %.4f | deep %s | whole solve %s' % (d['value'], (d['roofline'].get('kernel_ms') or 0), (d.get('deep_loop_state') or {}).get('value'), fs.get('iterations_per_s')))"; }
for rep in 1 2 3; do for lib in base lp; do
  [ $lib = base ] && unset EXAADMM_HIP_LIB || export EXAADMM_HIP_LIB=$LP
  timeout 300 python bench.py --no-cpu-baseline --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/h_${lib}_$rep.log 2>&1; line $O/h_${lib}_$rep.log "13659 default command $lib rep $rep"
done; done | tee $O/ab.txt
for lib in base lp; do
  [ $lib = base ] && unset EXAADMM_HIP_LIB || export EXAADMM_HIP_LIB=$LP
  timeout 300 python bench.py --case case_ACTIVSg70k --steps 100 --warmup 20 --state prep --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/k70_$lib.log 2>&1; line $O/k70_$lib.log "70k $lib"
  timeout 300 python bench.py --case case1354pegase --steps 100 --warmup 20 --no-cpu-baseline --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/c1354_$lib.log 2>&1; line $O/c1354_$lib.log "1354 $lib"
done | tee -a $O/ab.txt
