O=gpurun_out/r02_b12; mkdir -p $O
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve > $R/$O/trace20.log 2>&1
cd $R
python scripts/kernel_gaps.py $(ls $O/trace20/*/*kernel_trace.csv) 100
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('K=20: %.1f it/s ms/step %.4f device_ms_total %.4f' % (d['value'], d['ms_per_step'], d['solver']['device_ms_total']))
"; done
