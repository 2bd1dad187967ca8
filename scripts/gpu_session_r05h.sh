O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/resident.txt 2>&1; tail -3 $O/resident.txt
bash scripts/gpu_ab.sh $O "base default" 3 2>&1 | tee $O/ab.txt
bash scripts/gpu_ab.sh $O/c1354 "base default" 2 --case case1354pegase 2>&1 | tee $O/ab1354.txt
