O=gpurun_out/r05t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/resident.txt 2>&1; tail -4 $O/resident.txt
for a in 0 4 3; do for c in 50 0; do echo "adopt $a chunk $c"; EXAADMM_HEAVY_ADOPT=$a EXAADMM_RESIDENT_CHUNK=$c bash scripts/gpu_ab.sh $O/a${a}c$c "default" 2 2>&1; done; done | tee $O/ab.txt
for a in 0 4; do for c in 50 0; do echo "1354 adopt $a chunk $c"; EXAADMM_HEAVY_ADOPT=$a EXAADMM_RESIDENT_CHUNK=$c bash scripts/gpu_ab.sh $O/e${a}c$c "default" 1 --case case1354pegase 2>&1; done; done | tee $O/ab1354.txt
