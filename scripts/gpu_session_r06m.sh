# round 6, session m: the -m gpu suite under every A/B switch that selects another code path (defaults must not be the only tested path)
bash scripts/run_gpu_suite_under_switches.sh gpurun_out/r06_switches 2>&1 | tee gpurun_out/r06_switches_summary.txt
