# round 6, session b: GPU suite after the option / launch-cap / agreement changes, the driver's command with the new 1M leg
O=gpurun_out/r06b; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/tests.txt 2>&1; tail -5 $O/tests.txt
grep -h "branches with other\|outer 2 iteration\|sampled branches" $O/tests.txt | tee $O/parity_counts.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; tail -c 1500 $O/bench_driver.log
