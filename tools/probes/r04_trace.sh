O=gpurun_out/r04_trace; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); ls -la $f; head -2 $f
python3 - $f $GRAFT_REPO_ROOT/$O/trace_small.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Queue_Id", "Stream_Id", "Start_Timestamp", "End_Timestamp", "Workgroup_Size", "Grid_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count"]
keep = [k for k in keep if k in rows[0]]
with open(sys.argv[2], "w") as o:
    w = csv.writer(o); w.writerow(keep)
    for r in rows[-3000:]:
        w.writerow([r[k][:60] if k == "Kernel_Name" else r[k] for k in keep])
print(len(rows), "rows", list(rows[0].keys()))
PY
