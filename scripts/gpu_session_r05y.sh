O=gpurun_out/r05y; mkdir -p $O
for c in 0 100 50 24; do for r in 1 2; do
  EXAADMM_RESIDENT_CHUNK=$c EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_chunk2.so timeout 300 python bench.py --case case_ACTIVSg70k --steps 200 --warmup 20 --no-cpu-baseline --no-north-star --no-full-solve --gpu-seconds 3 > $O/c${c}_$r.log 2>&1
  python3 -c "
import json
for l in open('$O/c${c}_$r.log'):
    if l.startswith('{'):
        d=json.loads(l); print('chunk $c rep $r: value %.0f it/s  kernel_us %.2f' % (d['value'], 1e3*(d['roofline'].get('kernel_ms') or 0)))"
done; done | tee $O/ab.txt
