import csv,glob,sys,os
agg={}
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "dmm" not in k: continue
        e=agg.setdefault(k,{})
        e[r["Counter_Name"]]=e.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
rows=sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0))
for k,c in rows[:22]:
    wc=max(c.get("SQ_WAVE_CYCLES",1),1)
    print("%-60s bankconf/ldsactive=%.3f ldsactive/wavecyc=%.3f conf/wavecyc=%.3f"%(k.replace('_ZN3dmm12','')[:60], c.get("SQ_LDS_BANK_CONFLICT",0)/max(c.get("SQ_ACTIVE_INST_LDS",1),1), c.get("SQ_ACTIVE_INST_LDS",0)/wc, c.get("SQ_LDS_BANK_CONFLICT",0)/wc))
