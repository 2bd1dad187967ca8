mkdir -p gpurun_out/r02_conv
L=gpurun_out/r02_conv/sweep2.log
: > $L
python scripts/exp_convergence.py case1354pegase "dict()" 20 60 >> $L 2>&1
python scripts/exp_convergence.py case13659pegase "dict()" 20 90 >> $L 2>&1
python scripts/exp_convergence.py case13659pegase "dict(seed=1)" 20 90 >> $L 2>&1
python scripts/exp_convergence.py case_ACTIVSg70k "dict(scale=1e-5, grid=dict(cost='market'))" 20 120 >> $L 2>&1
python scripts/exp_convergence.py case_ACTIVSg70k "dict(scale=1e-4, rho=(5e1,5e3))" 20 120 >> $L 2>&1
python scripts/exp_convergence.py case118 "dict()" 25 30 >> $L 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02_conv/pytest2.log 2>&1
tail -5 gpurun_out/r02_conv/pytest2.log
