O=gpurun_out/r05p; mkdir -p $O
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 30 --no-cpu-baseline --no-full-solve --no-north-star > $R/$O/bench20.log 2>&1
cd $R
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
# the last 30 timed regions: find the last 30 launches of the heavy resident kernel and print what surrounds each
idx=[i for i,r in enumerate(rows) if "k_resident_t<1, true" in r["Kernel_Name"]]
print("heavy resident launches:", len(idx))
import collections
def short(n): return n.split("(")[0].replace("void exa::","")[:40]
for i in idx[-3:]:
    t0=int(rows[i]["Start_Timestamp"])
    for r in rows[max(0,i-6):i+6]:
        print("  %-42s start %+9.1f us  dur %8.1f us" % (short(r["Kernel_Name"]), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
    print()
d=[(int(rows[i]["End_Timestamp"])-int(rows[i]["Start_Timestamp"]))/1e3 for i in idx[-30:]]
print("median duration of the last 30 heavy launches: %.1f us" % sorted(d)[len(d)//2])
gaps=[(int(rows[idx[k+1]]["Start_Timestamp"])-int(rows[idx[k]]["Start_Timestamp"]))/1e3 for k in range(len(idx)-31,len(idx)-1)]
print("median start-to-start of consecutive regions: %.1f us" % sorted(gaps)[len(gaps)//2])
PY
rm -f $f
grep '^{' $O/bench20.log | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('value', d['value'], 'ms/step', d['ms_per_step'])"
