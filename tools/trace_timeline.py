#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per queue busy time, gaps, per-kernel in-step time, 1-ms bins.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o run -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile
    python3 tools/trace_timeline.py /tmp/tr/**/run_kernel_trace.csv

A step = the kernels between two consecutive adam kernels (the last complete step of the trace is used).  This is the measurement
behind DESIGN 4 "What a step is made of" (round 4): the main queue 92 % busy with a median gap of zero between dependent launches."""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("_ZN3dmm", "").replace("void ", "").replace("dmm::", "")
    m = re.match(r"\d*([A-Za-z0-9_]+?)_kernel", n)
    return m.group(1) if m else n[:20]


def main(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    if len(adam) < 2:
        sys.exit("fewer than two adam kernels in the trace")
    step = rows[adam[-2] + 1:adam[-1] + 1]
    t0, t1 = min(r["s"] for r in step), max(r["e"] for r in step)
    print(f"step: {(t1 - t0) / 1e6:.2f} ms wall, {len(step)} kernels, kernel time {sum(r['e'] - r['s'] for r in step) / 1e6:.2f} ms")
    byq = collections.defaultdict(list)
    for r in step:
        byq[r["Queue_Id"]].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        rs.sort(key=lambda r: r["s"])
        gaps = [rs[i + 1]["s"] - rs[i]["e"] for i in range(len(rs) - 1)] or [0]
        pos = sorted(g for g in gaps if g > 0)
        print(f"queue {q}: {len(rs)} kernels, busy {sum(r['e'] - r['s'] for r in rs) / 1e6:.2f} ms, gaps {sum(pos) / 1e6:.2f} ms "
              f"(median {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us, {sum(1 for g in gaps if g > 20000)} above 20 us, largest {max(gaps) / 1e3:.0f} us)")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        k = (r["Queue_Id"], short(r["Kernel_Name"]))
        agg[k][0] += 1
        agg[k][1] += (r["e"] - r["s"]) / 1e3
    print("in-step time by kernel:")
    for (q, k), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
        print(f"  q{q} {k:24s} n={n:4d} {t / 1e3:6.2f} ms  avg {t / n:7.1f} us")
    bins = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in step:
        s, e = (r["s"] - t0) / 1e6, (r["e"] - t0) / 1e6
        b = int(s)
        while b <= int(e):
            lo, hi = max(s, b), min(e, b + 1)
            if hi > lo:
                bins[b][r["Queue_Id"]] += hi - lo
            b += 1
    print("busy share per millisecond: " + " ".join(f"{b}:" + "/".join(f"{bins[b][q] * 100:.0f}" for q in sorted(byq)) for b in sorted(bins)))


if __name__ == "__main__":
    main(sys.argv[1])
