# round 6, session c: the whole GPU suite (no -x) after the option / agreement changes
O=gpurun_out/r06c; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -s > $O/tests.txt 2>&1; tail -8 $O/tests.txt
grep -h "branches with other\|outer 2 iteration\|sampled branches" $O/tests.txt | tee $O/parity_counts.txt
