"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel class.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [--config c2 --batch 4 --dtype f16]

Units/corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB; on gfx950
FETCH_SIZE tallies 128-B requests of 16-B-per-lane loads at 64 B, so reads are doubled; WRITE_SIZE is taken as is."""
import argparse
import csv
import glob
import json
import os
import re
import sys

EPI = {"0": "store", "1": "bnbwd", "2": "logits"}


def classify(name):
    m = re.search(r"igemm_kernelI(?:DF16_|DF16b|f)Li(\d+)ELi(\d+)E", name)
    if m:
        return f"igemm.{EPI.get(m.group(2), m.group(2))}.n{m.group(1)}"
    # conv3_kernel<T, CS, SPAN, TSPAN, TSTR, NT, GC, EPI, PRO>
    m = re.search(r"conv3_kernelI(?:DF16_|DF16b)Li\d+ELi\d+ELin?\d+ELi\d+ELi(\d+)ELi\d+ELi(\d+)ELi\d+E", name)
    if m:
        return f"conv3.{EPI.get(m.group(2), m.group(2))}.n{32 * int(m.group(1))}"
    if "pig_kernel" in name:
        return "pig.store.n128"
    if "thin_logits_kernel" in name:
        return "thin.logits.n32"
    if "wg3_kernel" in name:
        return "wg3.n128"
    if "wgp_kernel" in name:
        return "wgp"
    if "wg5_kernel" in name:
        return "wg5.n64"
    if "cvp_kernel" in name:
        return "cvp.store.n128"
    if "cvd_kernel" in name:
        return "cvp.bnbwd.n128"
    if "bw1_kernel" in name:
        return "bw1"
    m = re.search(r"halo_kernelI(?:DF16_|f)Li(\d+)ELi(\d+)E", name)
    if m:
        return f"igemm.{EPI.get(m.group(2), m.group(2))}.n{m.group(1)}"
    m = re.search(r"wgrad_kernelI(?:DF16_|DF16b|f)Li(\d+)E", name)
    if m:
        return f"wgrad*.n{m.group(1)}"
    m = re.search(r"dmm(?:::|\d+)(\w+?)_kernel", name)
    if m:
        return re.sub(r"^\d+", "", m.group(1))
    return None


def collect(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            cls = classify(r["Kernel_Name"])
            if cls is None:
                continue
            e = out.setdefault(cls, [0, 0.0])
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"), ap.add_argument("write_dir"), ap.add_argument("out")
    ap.add_argument("--config", default="c2"), ap.add_argument("--batch", type=int, default=4), ap.add_argument("--dtype", default="f16")
    a = ap.parse_args()
    fe, wr = collect(a.fetch_dir, "FETCH_SIZE"), collect(a.write_dir, "WRITE_SIZE")
    res = {}
    for cls in sorted(set(fe) | set(wr)):
        nf, kf = fe.get(cls, [0, 0.0])
        nw, kw = wr.get(cls, [0, 0.0])
        rd = 2.0 * 1024.0 * kf / nf if nf else None
        wb = 1024.0 * kw / nw if nw else None
        res[cls] = dict(launches_fetch_pass=nf, launches_write_pass=nw, read_bytes_per_launch=rd, write_bytes_per_launch=wb,
                        traffic_bytes_per_launch=(rd or 0) + (wb or 0))
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.src_hash import source_hash
    json.dump(dict(config=a.config, batch=a.batch, dtype=a.dtype, source_sha16=source_hash(),
                   method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB*1024; FETCH_SIZE x2 (gfx950)",
                   classes=res), open(a.out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * max(kv[1]["launches_fetch_pass"], 1)):
        print(f"{k:24s} n={v['launches_fetch_pass']:5d} rd/launch={(v['read_bytes_per_launch'] or 0)/1e6:9.2f} MB wr/launch={(v['write_bytes_per_launch'] or 0)/1e6:9.2f} MB")


if __name__ == "__main__":
    sys.exit(main())
