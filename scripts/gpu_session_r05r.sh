O=gpurun_out/r05r; mkdir -p $O
for w in 32 128 256 1000; do for c in 0 50; do echo "window $w chunk $c"; EXAADMM_HEAVY_WINDOW=$w EXAADMM_RESIDENT_CHUNK=$c bash scripts/gpu_ab.sh $O/w${w}c$c "default" 2 2>&1; done; done | tee $O/abwin.txt
for w in 128 1000; do for c in 0 50; do echo "1354 window $w chunk $c"; EXAADMM_HEAVY_WINDOW=$w EXAADMM_RESIDENT_CHUNK=$c bash scripts/gpu_ab.sh $O/e${w}c$c "default" 1 --case case1354pegase 2>&1; done; done | tee $O/abwin1354.txt
