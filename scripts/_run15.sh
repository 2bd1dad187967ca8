O=gpurun_out/r02_b15; mkdir -p $O
python bench.py --case case_ACTIVSg70k --steps 100 --warmup 20 --cpu-solve-budget 200 > $O/bench_70k_full.log 2>&1
python bench.py --case case1354pegase --steps 100 --warmup 20 > $O/bench_1354_full.log 2>&1
for f in $O/*.log; do python -c "
import json
for l in open('$f'):
    if l.startswith('{'):
        d=json.loads(l); fs=d['full_solve']; print('$f', d['value'], {k: fs.get(k) for k in ('status','outer','cumul','iterations_per_s','speedup_vs_cpu_port','cpu_port_same_iteration_counts','objective_rel_diff_vs_cpu_port')}, fs['cpu_port']['sample'], d['cpu_baseline']['value'])
"; done
