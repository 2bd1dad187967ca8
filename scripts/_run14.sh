O=gpurun_out/r02_final2; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1
python bench.py > $O/bench_default.log 2>&1
for c in case1354pegase case_ACTIVSg70k synthetic1M; do python bench.py --case $c --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_$c.log 2>&1; done
python bench.py --force-dist --steps 200 --warmup 20 > $O/bench_dist1.log 2>&1
python bench.py --periods 4 --steps 100 --warmup 20 > $O/bench_mp4.log 2>&1
for f in $O/bench_*.log; do echo "$f: $(python -c "
import json
for l in open('$f'):
    if l.startswith('{'):
        d=json.loads(l); fs=d.get('full_solve',{}); cb=d.get('cpu_baseline',{}); print('%.1f it/s  ms/step %.4f k_branch %s | full solve %s | cpu %s speedup %s' % (d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_ms'), (fs.get('status'), fs.get('outer'), fs.get('cumul'), fs.get('iterations_per_s'), fs.get('speedup_vs_cpu_port')), cb.get('value'), d.get('speedup_vs_cpu_baseline')))
")"; done
