O=gpurun_out/r02_b13; mkdir -p $O
python scripts/iter_profile.py case13659pegase 600 > $O/iter_profile_13659.txt 2>&1
python scripts/iter_profile.py case_ACTIVSg70k 400 > $O/iter_profile_70k.txt 2>&1
cat $O/iter_profile_13659.txt
