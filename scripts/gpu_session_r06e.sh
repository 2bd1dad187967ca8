# round 6, session e: the two-wavefront form with ONE bulk fetch of the constants per trip (w2b) against lazy loads (w2) and the shipped library, 1M branches
O=gpurun_out/r06e; mkdir -p $O
line() { python3 -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); print('$2: value %.0f it/s  kernel_ms %.4f frac %.3f' % (d['value'], (d['roofline'].get('kernel_ms') or 0), d['roofline'].get('frac') or 0))"; }
for lib in base w2 w2b; do
  [ $lib = base ] && unset EXAADMM_HIP_LIB || export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_$lib.so
  timeout 300 python bench.py --case synthetic1M --state prep --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/m1_$lib.log 2>&1; line $O/m1_$lib.log "1M $lib"
done | tee $O/ab.txt
