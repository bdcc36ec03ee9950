#!/bin/bash
# tools/sweep_lib.sh <out-tag> <lib...>: bench (C2) with each experiment build build_var/lib_<name>.so ("main" = the in-tree library);
# prints the step time and the serial-pass time of the classes named in $CLASSES
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
for v in "$@"; do
  if [ "$v" = main ]; then unset DMM_LIB_PATH; else export DMM_LIB_PATH=$PWD/build_var/lib_$v.so; fi
  timeout -k 10 200 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --table --ops 2000 ${AB_ARGS} > $out/bench_$v.json 2> $out/bench_$v.txt || { echo "$v FAILED"; tail -3 $out/bench_$v.txt; continue; }
  python3 - "$v" $out/bench_$v.json $out/bench_$v.txt <<'PY'
import json, sys, os
tag, j, t = sys.argv[1:4]
d = json.load(open(j))
want = os.environ.get("CLASSES", "bw1.n128,wg3.n128,igemm.store.n128,other").split(",")
rows = {}
for l in open(t):
    if l.startswith('{"kernel"'):
        r = json.loads(l); rows[r["kernel"]] = r["ms_total"]
print(f"{tag:10s} step {d['ms_per_step']:7.3f}", " ".join(f"{k}={rows.get(k)}" for k in want), "serial_sum", round(sum(rows.values()), 2), flush=True)
PY
done
