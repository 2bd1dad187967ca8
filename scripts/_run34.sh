O=gpurun_out/r02_b34; mkdir -p $O
EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_prof.so python scripts/wave_timeline.py case13659pegase 200 2>/dev/null | grep "in-kernel stamps" | head -2 | cut -c1-250
for i in 1 2; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('13659: %.1f it/s k_branch %.4f full solve %.1f' % (d['value'], d['roofline']['kernel_ms'], d['full_solve']['iterations_per_s']))
"; done
for c in case1354pegase case_ACTIVSg70k; do python bench.py --case $c --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$c: %.1f it/s k_branch %.4f full solve %.1f' % (d['value'], d['roofline']['kernel_ms'], d['full_solve']['iterations_per_s']))
"; done
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
