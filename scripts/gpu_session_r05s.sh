O=gpurun_out/r05s; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; grep -n "passed\|failed" $O/suite.txt | tail -2
bash scripts/gpu_ab.sh $O/def "default" 2 2>&1 | tee $O/abdef.txt
bash scripts/gpu_ab.sh $O/def20 "default" 2 --steps 20 --warmup 5 2>&1 | tee $O/abdef20.txt
bash scripts/gpu_ab.sh $O/def1354 "default" 2 --case case1354pegase 2>&1 | tee $O/abdef1354.txt
