O=gpurun_out/r06s; mkdir -p $O
for sw in EXAADMM_HEAVY_SPLIT=0 EXAADMM_NO_HEAVY=1 EXAADMM_DIST_P2P=0 EXAADMM_RESIDENT_TILES=0 EXAADMM_NO_GRAPH=1; do
  env $sw timeout 1200 python -m pytest tests/test_gpu_dist_threads.py tests/test_gpu_resident.py tests/test_gpu_p2p_ipc.py -m gpu -q -p no:cacheprovider > $O/pytest_$sw.log 2>&1
  echo "$sw: $(grep -E 'passed|failed' $O/pytest_$sw.log | tail -1); retries $(grep -c 'one retry' $O/pytest_$sw.log)" | tee -a $O/summary.txt
done
