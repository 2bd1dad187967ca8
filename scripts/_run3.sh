O=gpurun_out/r02_b3
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python bench.py > $O/bench_default.log 2>&1
for v in "" _w2 _w2c; do
  for c in case13659pegase case_ACTIVSg70k synthetic1M; do
    EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip$v.so python bench.py --case $c --steps 100 --warmup 20 --no-cpu-baseline --no-full-solve > $O/bench${v}_$c.log 2>&1
    echo "variant '$v' $c: $(python -c "
import json,sys
for l in open('$O/bench${v}_$c.log'):
    if l.startswith('{'):
        d=json.loads(l); print('%.1f it/s  k_branch %.4f ms  occ %s' % (d['value'], d['roofline']['kernel_ms'], d['roofline'].get('waves_per_simd')))
")"
  done
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-solve > $R/$O/trace_bench.log 2>&1
cd $R
ls -la $O/trace/*/* | head
