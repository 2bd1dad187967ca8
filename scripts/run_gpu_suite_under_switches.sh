#!/bin/bash
# GPU box: the whole -m gpu suite under every A/B switch of the library (defaults must not be the only tested path).
# usage: bash scripts/run_gpu_suite_under_switches.sh [outdir]
O=${1:-gpurun_out/switches}; mkdir -p $O
for sw in ${SWITCHES:-"" EXAADMM_NO_RESIDENT=1 EXAADMM_NO_REC=1 EXAADMM_NO_SPEC=1 EXAADMM_NO_GRAPH=1 EXAADMM_BUS_KERNEL=lane EXAADMM_DIST_PYTHON_LOOP=1 EXAADMM_RESIDENT_TILES=0 EXAADMM_DIST_OVERLAP=0 EXAADMM_NO_HEAVY=1 EXAADMM_HEAVY_SPLIT=0 EXAADMM_BLOCK_ORDER=0 EXAADMM_DIST_P2P=0}; do
  tag=${sw:-default}
  env $sw timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_$tag.log 2>&1
  echo "$tag: $(grep -E 'passed|failed' $O/pytest_$tag.log | tail -1)"
done
