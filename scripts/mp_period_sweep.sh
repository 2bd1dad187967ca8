for T in 2 3 4 6; do timeout 300 python bench.py --periods $T --steps 100 --warmup 20 --no-cpu-baseline --gpu-seconds 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('periods $T: %.0f it/s, ms/step %.4f, k_branch %.4f ms, waves/SIMD %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['waves_launched_per_simd']))"; done
