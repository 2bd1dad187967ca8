# round 6, session d: two wavefronts per SIMD for the branch launch (libexaadmm_hip_w2.so, -DEXA_BR_WAVES=2) against the shipped library:
# parity of the two-launch loop's kernels under the variant, then interleaved A/B at 1M branches, 70k-sized (two-launch loop) and 4 periods
O=gpurun_out/r06d; mkdir -p $O
W2=$PWD/exaadmm.jl_amd/libexaadmm_hip_w2.so
EXAADMM_HIP_LIB=$W2 EXAADMM_NO_RESIDENT=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -s > $O/tests_w2.txt 2>&1; tail -4 $O/tests_w2.txt
grep -h "branches with other\|outer 2 iteration\|sampled branches" $O/tests_w2.txt | tee $O/parity_counts_w2.txt
line() { python3 -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); print('$2: value %.0f it/s  kernel_ms %.4f frac %.3f' % (d['value'], (d['roofline'].get('kernel_ms') or 0), d['roofline'].get('frac') or 0))"; }
for rep in 1 2; do for lib in base w2; do
  [ $lib = w2 ] && export EXAADMM_HIP_LIB=$W2 || unset EXAADMM_HIP_LIB
  timeout 300 python bench.py --case synthetic1M --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/m1_${lib}_$rep.log 2>&1; line $O/m1_${lib}_$rep.log "1M $lib rep $rep"
  EXAADMM_NO_RESIDENT=1 timeout 300 python bench.py --case case_ACTIVSg70k --steps 100 --warmup 20 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/k70_${lib}_$rep.log 2>&1; line $O/k70_${lib}_$rep.log "70k two-launch $lib rep $rep"
done; done | tee $O/ab.txt
for lib in base w2; do
  [ $lib = w2 ] && export EXAADMM_HIP_LIB=$W2 || unset EXAADMM_HIP_LIB
  timeout 300 python bench.py --periods 4 --steps 100 --warmup 20 --no-cpu-baseline --gpu-seconds 3 > $O/mp4_$lib.log 2>&1; line $O/mp4_$lib.log "4 periods $lib"
  EXAADMM_NO_RESIDENT=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/h_$lib.log 2>&1; line $O/h_$lib.log "13659 two-launch $lib"
done | tee -a $O/ab.txt
unset EXAADMM_HIP_LIB
timeout 400 python bench.py --steps 20 --warmup 5 --no-throughput-leg > $O/bench_driver_base.log 2>&1; line $O/bench_driver_base.log "driver command, shipped library (pow outlined)" | tee -a $O/ab.txt
