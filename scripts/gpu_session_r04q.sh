R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04q
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04q/trace -- python3 $R/bench.py --case case_ACTIVSg70k --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline --no-full-solve > $R/gpurun_out/r04q/bench.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04q/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last 3 resident launches of 100 iterations and print everything from 400 us before the first of them
res = [i for i, r in enumerate(rows) if "k_resident_t" in r["Kernel_Name"]]
i0 = res[-4]
t0 = int(rows[i0 - 8]["Start_Timestamp"])
prev_end = t0
for r in rows[i0 - 8:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f us  dur %9.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:70]))
    prev_end = e
PY
EXAADMM_RESIDENT_COOP=0 timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 2 --case case_ACTIVSg70k --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('coop=0 steps=100 value %.0f whole %.0f' % (r['value'], r['value_reference_metric']))"
timeout 600 python bench.py --no-cpu-baseline --gpu-seconds 2 --case case_ACTIVSg70k --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('coop=1 steps=100 value %.0f whole %.0f' % (r['value'], r['value_reference_metric']))"
find gpurun_out/r04q -name "*kernel_trace.csv" -delete
