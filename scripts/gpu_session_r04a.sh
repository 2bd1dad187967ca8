set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 1500 python -m pytest tests/test_gpu_resident.py -x -q -s > gpurun_out/r04a/resident_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04a/resident_tests.log
tail -5 gpurun_out/r04a/resident_tests.log
for v in 0 4 1 2 3; do timeout 300 scripts/microbench/mp_litmus $v; done > gpurun_out/r04a/litmus.txt 2>&1
cat gpurun_out/r04a/litmus.txt
for coop in 1 0; do
  for rep in 1 2; do
    EXAADMM_RESIDENT_COOP=$coop timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04a/bench_coop${coop}_$rep.json 2> gpurun_out/r04a/bench_coop${coop}_$rep.err
    python - <<PY
import json
r=json.loads(open("gpurun_out/r04a/bench_coop${coop}_$rep.json").read().strip().splitlines()[-1])
print("coop=$coop rep=$rep value", r["value"], "ref_metric", r.get("value_reference_metric"), "kernel_ms", r["roofline"]["kernel_ms"], r["roofline"]["loop"])
PY
  done
done
