# round 6, session h: GPU suite on the shipped library (pow outlined, launch cap, agreement, options), the driver's command with the on-trajectory state + legs
O=gpurun_out/r06h; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -s > $O/tests.txt 2>&1; tail -8 $O/tests.txt | cut -c1-300
grep -h "branches with other\|outer 2 iteration\|sampled branches\|reference-arithmetic build" $O/tests.txt | tee $O/parity_counts.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; tail -c 3000 $O/bench_driver.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -c 300 $O/bench_default.log
