O=gpurun_out/r02_b27; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for i in 1 2; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default_$i.log 2>&1; done
for c in case_ACTIVSg70k synthetic1M; do python bench.py --case $c --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_$c.log 2>&1; done
EXAADMM_BUS_KERNEL=lane python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-solve > $O/bench_buslane.log 2>&1
for f in $O/bench_*.log; do echo "$f: $(python -c "
import json
for l in open('$f'):
    if l.startswith('{'):
        d=json.loads(l); fs=d.get('full_solve',{}); print('%.1f it/s  ms/step %.4f k_branch %.4f | full solve %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], (fs.get('cumul'), fs.get('iterations_per_s'))))
")"; done
