# round 6, session f: why the two-wavefront form loses -- SQ / TCC counters of k_branch at 1M branches, shipped library against w2b
O=gpurun_out/r06f; mkdir -p $O
export PMC_GROUPS="FETCH_SIZE;GRBM_GUI_ACTIVE;SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY;SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_IFETCH;SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS;SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"
for lib in base w2b; do
  [ $lib = base ] && unset EXAADMM_HIP_LIB || export EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_$lib.so
  bash scripts/collect_pmc.sh synthetic1M $O/pmc_$lib 20 --state prep | tee $O/passes_$lib.txt
  python3 scripts/summarize_pmc.py $O/pmc_$lib synthetic1M-$lib 20 > $O/summary_$lib.json 2>$O/summary_$lib.err
  find $O/pmc_$lib -name "*.csv" -size +2M -delete; find $O/pmc_$lib -name "*kernel_trace.csv" -delete
done
cat $O/summary_base.json $O/summary_w2b.json
