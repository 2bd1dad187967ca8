# round 6, session a: baseline of the round-5 tree on today's box (GPU suite, headline line, 1M and 70k legs, 4 periods)
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; tail -c 600 $O/bench_driver.log
timeout 300 python bench.py --case synthetic1M --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve --no-north-star --gpu-seconds 3 > $O/m1.log 2>&1
timeout 300 python bench.py --case case_ACTIVSg70k --steps 100 --warmup 20 --no-cpu-baseline --no-full-solve --no-north-star --gpu-seconds 3 > $O/k70.log 2>&1
timeout 300 python bench.py --periods 4 --steps 100 --warmup 20 --no-cpu-baseline --gpu-seconds 3 > $O/mp4.log 2>&1
for f in m1 k70 mp4; do python3 -c "
import json
for l in open('$O/$f.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$f: value %.0f it/s  kernel_ms %.4f frac %.3f' % (d['value'], (d['roofline'].get('kernel_ms') or 0), d['roofline'].get('frac') or 0))"; done | tee $O/summary.txt
