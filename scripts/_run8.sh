O=gpurun_out/r02_b8
mkdir -p $O
python scripts/diag_stragglers.py case13659pegase 200 > $O/stragglers_13659.txt 2>&1
python scripts/diag_stragglers.py case_ACTIVSg70k 200 > $O/stragglers_70k.txt 2>&1
EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_prof.so python scripts/wave_timeline.py case13659pegase 200 > $O/wave_timeline_13659.txt 2>&1
cat $O/stragglers_13659.txt; tail -30 $O/wave_timeline_13659.txt
