import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
adam=[i for i,r in enumerate(rows) if 'adam' in r['Kernel_Name']]
i0=adam[-2]; t0=rows[i0]['s']
sel=sorted(rows[i0:i0+int(sys.argv[2])], key=lambda r:r['s'])
for r in sel:
    print(f"q{r['Queue_Id']} {(r['s']-t0)/1e3:9.1f} -> {(r['e']-t0)/1e3:9.1f} us  {r['Kernel_Name'].replace('_ZN3dmm','').replace('void ','')[:56]}")
