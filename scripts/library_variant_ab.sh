#!/bin/bash
# GPU box: A/B of library VARIANTS built next to the shipped one (make -C exaadmm.jl_amd/csrc EXTRA="-D..." OUT=../libexaadmm_hip_<tag>.so variant;
# diagnostic builds add -DEXA_PROFILE_PHASES) -- how profiles/archive_r03_r04/r04_lone_search.txt was measured.
#   library_variant_ab.sh bench   <tag>...   headline bench line per variant, with and without EXAADMM_NO_LONE_SEARCH=1
#   library_variant_ab.sh phases  <tag>...   per-phase clocks of lone-lane solver trips (scripts/phase_cycles.py; diagnostic builds)
#   library_variant_ab.sh icache  <tag>...   instruction-cache / fetch counters of the dominant kernel (rocprofv3 --pmc)
mode=$1; shift
R=$PWD; O=$R/gpurun_out/variants; mkdir -p $O
for v in "$@"; do
  lib=$R/exaadmm.jl_amd/libexaadmm_hip_$v.so
  for sw in "" EXAADMM_NO_LONE_SEARCH=1; do
    tag=${v}${sw:+_serial}
    case $mode in
    bench)
      env $sw EXAADMM_HIP_LIB=$lib timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-solve > $O/bench_$tag.log 2>&1
      python3 -c "
import json
for l in open('$O/bench_$tag.log'):
    if l.startswith('{'): print('$tag: %.0f it/s' % json.loads(l)['value'])" ;;
    phases)
      echo "=== $tag"
      env $sw EXAADMM_HIP_LIB=$lib EXAADMM_NO_RESIDENT=1 timeout 200 python scripts/phase_cycles.py 300 2>&1 | grep -v amdgpu.ids | tail -4 ;;
    icache)
      [ -n "$sw" ] || continue
      cd /tmp; export TMPDIR=/tmp
      for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
        d=$O/pmc_$tag
        env $sw EXAADMM_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-full-solve > $d.log 2>&1
        python3 - "$(find $d -name '*counter_collection.csv' | head -1)" "$tag" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k in acc:
    if 'resident' in k or 'branch' in k: print(sys.argv[2],k,{c:'%.3g'%(v/n[(k,c)]) for c,v in acc[k].items()})
PY
        rm -rf $d
      done
      cd $R ;;
    esac
  done
done
