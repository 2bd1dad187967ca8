# round 6: stress of the cross-wavefront hand-overs on the FINAL build (LDS parking changes every kernel's timing): resident loop, adoption, partitions
O=gpurun_out/r06q; mkdir -p $O
timeout 1200 python scripts/stress_resident.py 40 > $O/stress_resident.txt 2>&1; tail -4 $O/stress_resident.txt
timeout 1200 python scripts/stress_adoption.py 400 > $O/stress_adoption.txt 2>&1; tail -6 $O/stress_adoption.txt
HK=$PWD/exaadmm.jl_amd/libexaadmm_hip_hooks.so
( export EXAADMM_HIP_LIB=$HK EXAADMM_RCCL_LIB=tests/rccl_shim/librccl_shim.so GPU_MAX_HW_QUEUES=16 EXAADMM_RESIDENT_COOP=0 EXAADMM_HEAVY_MAX=100 EXAADMM_HEAVY_SELECT_SMALL=1
  timeout 900 python scripts/stress_p2p.py 2 case1354pegase 1 100
  timeout 900 python scripts/stress_p2p.py 3 case1354pegase 1 60
  timeout 900 python scripts/stress_p2p.py 2 case13659pegase 2 40 ) > $O/stress_p2p.txt 2>&1; grep -v "^\[" $O/stress_p2p.txt | tail -8
