O=gpurun_out/r02_b19; mkdir -p $O
EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_prof.so python scripts/wave_timeline.py case13659pegase 320 > $O/wave_timeline.txt 2>&1
EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_prof.so python scripts/phase_cycles.py > $O/phase_cycles.txt 2>&1
tail -22 $O/wave_timeline.txt; tail -30 $O/phase_cycles.txt
