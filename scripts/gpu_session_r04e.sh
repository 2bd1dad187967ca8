set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
export EXAADMM_RCCL_LIB=$PWD/tests/rccl_shim/librccl_shim.so RCCL_SHIM_ASYNC=1 GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04e/trace -- python3 $GRAFT_REPO_ROOT/scripts/dist_overlap_probe.py 2 case13659pegase 60 > $GRAFT_REPO_ROOT/gpurun_out/r04e/probe.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r04e/probe.log
find gpurun_out/r04e/trace -name "*kernel_trace.csv" | head
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r04e/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), rows[0].keys())
# keep the last 400 kernels, print stream/queue, name, start, end relative
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-700:-300]
t0 = int(tail[0]["Start_Timestamp"])
out = open("gpurun_out/r04e/timeline_tail.txt", "w")
for r in tail:
    name = r["Kernel_Name"][:60]
    out.write("%8.2f %8.2f q%s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), name))
out.close()
PY
head -80 gpurun_out/r04e/timeline_tail.txt
