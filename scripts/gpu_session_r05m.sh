O=gpurun_out/r05m; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; grep -n "passed\|failed" $O/suite.txt | tail -2
bash scripts/gpu_ab.sh $O/def "default" 2 2>&1
bash scripts/gpu_ab.sh $O/def20 "default" 2 --steps 20 --warmup 5 2>&1
