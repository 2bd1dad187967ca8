#!/bin/bash
# GPU box: interleaved A/B of library variants on ONE box.  usage: gpu_ab.sh OUTDIR "tagA tagB ..." REPS [bench args...]
# tag "default" = the shipped library, any other tag = exaadmm.jl_amd/libexaadmm_hip_<tag>.so
O=$1; TAGS=$2; REPS=$3; shift 3
mkdir -p $O
for r in $(seq 1 $REPS); do
  for t in $TAGS; do
    lib=$PWD/exaadmm.jl_amd/libexaadmm_hip_$t.so; [ "$t" = default ] && lib=$PWD/exaadmm.jl_amd/libexaadmm_hip.so
    EXAADMM_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-north-star --gpu-seconds 2 "$@" > $O/ab_${t}_$r.log 2>&1
    python3 -c "
import json
for l in open('$O/ab_${t}_$r.log'):
    if l.startswith('{'):
        d=json.loads(l); fs=d.get('full_solve') or {}
        print('$t rep $r: value %.0f it/s  kernel_us %.2f  whole %s' % (d['value'], 1e3*(d['roofline'].get('kernel_ms') or 0), fs.get('iterations_per_s')))"
  done
done
