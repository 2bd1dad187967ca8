O=gpurun_out/r05u; mkdir -p $O
export EXAADMM_RESIDENT_CHUNK=0
for a in 4 5 6; do echo "adopt $a"; EXAADMM_HEAVY_ADOPT=$a bash scripts/gpu_ab.sh $O/a$a "default" 2 2>&1; done | tee $O/ab.txt
for a in 4 8 12; do echo "heavy from the first launch, adopt $a"; EXAADMM_HEAVY_FROM=0 EXAADMM_HEAVY_SPLIT=0 EXAADMM_HEAVY_ADOPT=$a bash scripts/gpu_ab.sh $O/f$a "default" 2 2>&1; done | tee $O/abfrom0.txt
for a in 4 8; do echo "1354 heavy from the first launch, adopt $a"; EXAADMM_HEAVY_FROM=0 EXAADMM_HEAVY_SPLIT=0 EXAADMM_HEAVY_ADOPT=$a bash scripts/gpu_ab.sh $O/g$a "default" 1 --case case1354pegase 2>&1; done | tee $O/abfrom0_1354.txt
