# round 6: localising the rare alternative trajectory of the 3-rank case118 solve over the mailbox loop (heavy_split on partitions, new this round)
O=gpurun_out/r06r2; mkdir -p $O
run() { tag=$1; n=$2; shift 2; ok=0; bad=0
  for i in $(seq 1 $n); do
    if env "$@" timeout 300 python -m pytest "tests/test_gpu_dist_threads.py::test_resident_loop_with_peers_is_the_single_handle_solve" -m gpu -x -q -k "case118" > $O/${tag}_$i.log 2>&1; then ok=$((ok+1)); rm -f $O/${tag}_$i.log; else bad=$((bad+1)); fi
  done; echo "$tag: $ok passed, $bad failed of $n"; }
run adopt0 45 EXAADMM_HEAVY_ADOPT=0 | tee $O/summary.txt
run heavymax0_split_only 45 EXAADMM_TEST_HEAVY_MAX_OVERRIDE=0 | tee -a $O/summary.txt
