O=gpurun_out/r05q; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; grep -n "passed\|failed" $O/suite.txt | tail -2
bash scripts/gpu_ab.sh $O "default" 2 --steps 20 --warmup 5 2>&1 | tee $O/ab20.txt
bash scripts/gpu_ab.sh $O/k200 "default" 2 2>&1 | tee $O/ab200.txt
