#!/bin/bash
# GPU box: rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only; MI355X_MICROARCH.md "rocprofv3 PMC slots")
# for the bench command of one workload.  usage: collect_pmc.sh CASE OUTDIR [steps] [extra bench flags...]
# (PMC_GROUPS: ';'-separated counter groups instead of the default six.)  Summarise with scripts/summarize_pmc.py OUTDIR workload
# -> profiles/r06_pmc_summary.json (scripts/refresh_profiles.py)
CASE=${1:-case13659pegase}
OUT=${2:-gpurun_out/pmc_$CASE}
STEPS=${3:-200}
shift 3 2>/dev/null
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra GROUPS_ <<< "${PMC_GROUPS:-FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE;SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY;SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_IFETCH}"
for grp in "${GROUPS_[@]}"; do
  tag=$(echo $grp | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/$tag -- python3 $R/bench.py --case $CASE --steps $STEPS --warmup 20 --repeats 1 --no-cpu-baseline --no-full-solve --no-north-star --no-throughput-leg --no-deep-loop-leg "$@" > $R/$OUT/$tag.log 2>&1
  echo "$tag rc=$?"
done
cd $R
