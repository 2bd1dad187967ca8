O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/resident.txt 2>&1; tail -3 $O/resident.txt
bash scripts/gpu_ab.sh $O "nolook default" 3 2>&1 | tee $O/ab.txt
bash scripts/gpu_ab.sh $O/c1354 "nolook default" 2 --case case1354pegase 2>&1 | tee $O/ab1354.txt
EXAADMM_HIP_LIB=$PWD/exaadmm.jl_amd/libexaadmm_hip_series.so EXAADMM_PROF_RESIDENT=1 timeout 300 python scripts/heavy_series.py case13659pegase 300 40 2>&1 | grep -v amdgpu.ids > $O/heavy_series.txt; grep -A16 "LONGEST" $O/heavy_series.txt
