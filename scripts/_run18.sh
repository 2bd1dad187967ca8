O=gpurun_out/r02_b18; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default.log 2>&1
for c in case_ACTIVSg70k synthetic1M; do python bench.py --case $c --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_$c.log 2>&1; done
python bench.py --periods 4 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_mp4.log 2>&1
for f in $O/bench_*.log; do echo "$f: $(python -c "
import json
for l in open('$f'):
    if l.startswith('{'):
        d=json.loads(l); fs=d.get('full_solve',{}); print('%.1f it/s  ms/step %.4f k_branch %.4f | full solve %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], (fs.get('cumul'), fs.get('iterations_per_s'))))
")"; done
