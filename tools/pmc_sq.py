"""Summarise a rocprofv3 --pmc SQ_* pass per kernel: python tools/pmc_sq.py <dir> <out.csv>"""
import csv
import glob
import os
import sys


def main(d, out):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dmm" not in k:
                continue
            e = agg.setdefault(k, dict(disp=set(), wg=0, c={}))
            if r["Dispatch_Id"] not in e["disp"]:
                e["disp"].add(r["Dispatch_Id"])
                e["wg"] += int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = []
    for k, e in agg.items():
        c, wg = e["c"], max(e["wg"], 1)
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        rows.append([k, len(e["disp"]), e["wg"], round(c.get("SQ_WAIT_ANY", 0) / wc, 3), round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                     round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), round(c.get("SQ_INSTS_VALU", 0) / wg), round(c.get("SQ_INSTS_MFMA", 0) / wg),
                     round(c.get("SQ_INSTS_LDS", 0) / wg), round(c.get("SQ_INSTS_VMEM", 0) / wg), round(wc / wg)])
    rows.sort(key=lambda r: -r[2] * r[10])
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "workgroups", "WAIT_ANY/WAVE_CYCLES", "WAIT_INST_ANY/WAVE_CYCLES", "ACTIVE_INST_ANY/WAVE_CYCLES",
                    "VALU_per_WG", "MFMA_per_WG", "LDS_per_WG", "VMEM_per_WG", "WAVE_CYCLES_per_WG"])
        w.writerows(rows)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
