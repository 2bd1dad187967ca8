O=gpurun_out/r05m; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; grep -n "passed\|failed" $O/suite.txt | tail -2
