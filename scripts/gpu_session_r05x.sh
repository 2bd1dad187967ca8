O=gpurun_out/r05x; mkdir -p $O
for sp in 24 32 48; do for a in 4 6; do echo "split $sp adopt $a"; EXAADMM_HEAVY_SPLIT=$sp EXAADMM_HEAVY_ADOPT=$a bash scripts/gpu_ab.sh $O/s${sp}a$a "default" 2 --steps 20 --warmup 5 2>&1; done; done | tee $O/ab.txt
for sp in 24 32 48; do for a in 4 6; do echo "1354 split $sp adopt $a"; EXAADMM_HEAVY_SPLIT=$sp EXAADMM_HEAVY_ADOPT=$a bash scripts/gpu_ab.sh $O/e${sp}a$a "default" 2 --case case1354pegase --steps 20 --warmup 5 2>&1; done; done | tee $O/ab1354.txt
