"""The first N (or, with a negative N, the last -N) kernels of the last complete step of a rocprofv3 kernel trace, by hardware queue:
   python3 tools/probes/dump_first.py <kernel_trace.csv> <N>   (a step = adam kernel to adam kernel; times relative to its start)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
i0, i1 = adam[-2], adam[-1]
t0 = rows[i0]['s']
n = int(sys.argv[2])
sel = rows[i0:i0 + n] if n > 0 else rows[i1 + n:i1 + 1]
for r in sel:
    print(f"q{r['Queue_Id']} {(r['s'] - t0) / 1e3:9.1f} -> {(r['e'] - t0) / 1e3:9.1f} us  {r['Kernel_Name'].replace('_ZN3dmm', '').replace('void ', '')[:56]}")
