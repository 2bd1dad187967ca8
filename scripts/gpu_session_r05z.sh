O=gpurun_out/r05z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mpacopf.py tests/test_gpu_scale.py -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for bo in 0 1; do for r in 1 2; do
  EXAADMM_BLOCK_ORDER=$bo timeout 300 python bench.py --periods 4 --steps 100 --warmup 20 --no-cpu-baseline --gpu-seconds 3 > $O/mp4_${bo}_$r.log 2>&1
  python3 -c "
import json
for l in open('$O/mp4_${bo}_$r.log'):
    if l.startswith('{'):
        d=json.loads(l); print('mp4 block_order $bo rep $r: value %.0f it/s  kernel_us %.2f' % (d['value'], 1e3*(d['roofline'].get('kernel_ms') or 0)))"
done; done | tee $O/ab_mp4.txt
for bo in 0 1; do
  EXAADMM_BLOCK_ORDER=$bo timeout 300 python bench.py --case synthetic1M --steps 50 --warmup 10 --no-cpu-baseline --no-full-solve --no-north-star --gpu-seconds 3 > $O/m1_${bo}.log 2>&1
  python3 -c "
import json
for l in open('$O/m1_${bo}.log'):
    if l.startswith('{'):
        d=json.loads(l); print('1M block_order $bo: value %.0f it/s  kernel_us %.2f' % (d['value'], 1e3*(d['roofline'].get('kernel_ms') or 0)))"
  EXAADMM_RESIDENT_TILES=0 EXAADMM_BLOCK_ORDER=$bo timeout 300 python bench.py --case case_ACTIVSg70k --steps 100 --warmup 20 --no-cpu-baseline --no-full-solve --no-north-star --gpu-seconds 3 > $O/k70_${bo}.log 2>&1
  python3 -c "
import json
for l in open('$O/k70_${bo}.log'):
    if l.startswith('{'):
        d=json.loads(l); print('70k two-launch block_order $bo: value %.0f it/s  kernel_us %.2f' % (d['value'], 1e3*(d['roofline'].get('kernel_ms') or 0)))"
done | tee $O/ab_other.txt
