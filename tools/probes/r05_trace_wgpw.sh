#!/bin/bash
# round 5: real (rocprofv3 kernel-trace) durations of the wgpw dispatches inside the step
out=gpurun_out/r05_trace_wgpw; mkdir -p $out; export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/ktr && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $out/prof.log 2>&1 || exit 2
f=$(find /tmp/ktr -name "*kernel_trace.csv" | head -1)
python3 - $f > $out/wgpw_dispatches.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "wgpw" in r["Kernel_Name"]]
n = len(sel) // 5
for r in sel[-n:]:
    print("%-60s %8.1f us grid %s wg %s lds %s" % (r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("LDS_Block_Size", "")))
PY
cat $out/wgpw_dispatches.txt
