# round 6, session k: GPU suite on the build with LDS parking, then the final batch (profiles/r06_*)
O=gpurun_out/r06k; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -s > $O/tests.txt 2>&1; tail -6 $O/tests.txt | cut -c1-300
grep -h "branches with other\|outer 2 iteration\|sampled branches\|reference-arithmetic build" $O/tests.txt | tee $O/parity_counts.txt
BUILD_NOTE="round 6 final build (LDS parking, pow outlined, launch cap, 15 options)" bash scripts/final_batch.sh gpurun_out/r06_final r06 2>&1 | tail -20
