# round 6, session i: persistent wavefronts for launches with more blocks than SIMDs (option block_order = 2) -- bitwise test, A/B against block_order = 1
O=gpurun_out/r06i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_resident.py -m gpu -x -q -k "persistent" > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
line() { python3 -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); print('$2: value %.0f it/s  kernel_ms %.4f frac %.3f' % (d['value'], (d['roofline'].get('kernel_ms') or 0), d['roofline'].get('frac') or 0))"; }
for rep in 1 2; do for bo in 1 2; do
  EXAADMM_BLOCK_ORDER=$bo timeout 300 python bench.py --case synthetic1M --state prep --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/m1_${bo}_$rep.log 2>&1; line $O/m1_${bo}_$rep.log "1M block_order $bo rep $rep"
  EXAADMM_BLOCK_ORDER=$bo timeout 300 python bench.py --periods 4 --steps 100 --warmup 20 --no-cpu-baseline --gpu-seconds 3 > $O/mp4_${bo}_$rep.log 2>&1; line $O/mp4_${bo}_$rep.log "4 periods block_order $bo rep $rep"
  EXAADMM_BLOCK_ORDER=$bo EXAADMM_RESIDENT_TILES=0 timeout 300 python bench.py --case case_ACTIVSg70k --state prep --steps 100 --warmup 20 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --gpu-seconds 3 > $O/k70_${bo}_$rep.log 2>&1; line $O/k70_${bo}_$rep.log "70k two-launch block_order $bo rep $rep"
done; done | tee $O/ab.txt
