O=gpurun_out/r02_b6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -x > $O/pytest_scale_tp.log 2>&1; tail -3 $O/pytest_scale_tp.log
for c in case_ACTIVSg70k synthetic1M; do
  for tp in 0 1025; do
    EXAADMM_TP_MIN_BLOCKS=$tp python bench.py --case $c --steps 100 --warmup 20 --no-cpu-baseline --no-full-solve > $O/bench_tp${tp}_$c.log 2>&1
    echo "tp_min=$tp $c: $(python -c "
import json
for l in open('$O/bench_tp${tp}_$c.log'):
    if l.startswith('{'):
        d=json.loads(l); print('%.1f it/s  k_branch %.4f ms occ %s' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['waves_per_simd']))
")"
  done
done
