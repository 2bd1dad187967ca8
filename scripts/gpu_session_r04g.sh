cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
timeout 1500 python -m pytest tests/test_gpu_resident.py -x -q > gpurun_out/r04g/resident_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04g/resident_tests.log
tail -15 gpurun_out/r04g/resident_tests.log
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 1 > gpurun_out/r04g/bench_13659_$rep.json 2> gpurun_out/r04g/bench_13659_$rep.err
  timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 1 --case case_ACTIVSg70k > gpurun_out/r04g/bench_70k_$rep.json 2> gpurun_out/r04g/bench_70k_$rep.err
  EXAADMM_RESIDENT_TILES=0 timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 1 --case case_ACTIVSg70k > gpurun_out/r04g/bench_70k_two_launch_$rep.json 2> gpurun_out/r04g/bench_70k_two_launch_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04g/bench_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.0f" % r["value"], "whole solve %.0f" % r.get("value_reference_metric", 0), "kernel_us %.2f" % (1e3 * r["roofline"]["kernel_ms"]), r["roofline"]["loop"], r["full_solve"]["outer"], r["full_solve"]["cumul"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-500:])
PY
