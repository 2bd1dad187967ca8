O=gpurun_out/r05v; mkdir -p $O
timeout 1500 python scripts/stress_adoption.py 400 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/stress_adoption.txt
