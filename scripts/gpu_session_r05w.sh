O=gpurun_out/r05w; mkdir -p $O
for t in base default; do lib=$PWD/exaadmm.jl_amd/libexaadmm_hip_$t.so; [ $t = default ] && lib=$PWD/exaadmm.jl_amd/libexaadmm_hip.so; for r in 1 2 3; do
  EXAADMM_HIP_LIB=$lib timeout 300 python bench.py --case case_ACTIVSg70k --steps 100 --warmup 20 --no-cpu-baseline --no-north-star --gpu-seconds 3 > $O/ab70k_${t}_$r.log 2>&1
  python3 -c "
import json
for l in open('$O/ab70k_${t}_$r.log'):
    if l.startswith('{'):
        d=json.loads(l); fs=d.get('full_solve') or {}
        print('$t rep $r: value %.0f it/s  kernel_us %.2f  whole %s' % (d['value'], 1e3*(d['roofline'].get('kernel_ms') or 0), fs.get('iterations_per_s')))"
done; done | tee $O/ab70k.txt
SWITCHES="EXAADMM_NO_HEAVY=1 EXAADMM_DIST_P2P=0" bash scripts/run_gpu_suite_under_switches.sh gpurun_out/switches_r05c
