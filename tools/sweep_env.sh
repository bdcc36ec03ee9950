#!/bin/bash
# tools/sweep_env.sh <out-tag> <VAR> <value...>: bench (C2) under each value of an environment knob ("-" = unset); prints the step time
# and the serial-pass time of the classes named in $CLASSES (default: all of bw1 / wg3 / igemm.store)
tag=$1; var=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out
for v in "$@"; do
  if [ "$v" = - ]; then unset $var; else export $var=$v; fi
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --table ${AB_ARGS} > $out/bench_$v.json 2> $out/bench_$v.txt || exit 1
  python3 - "$var=$v" $out/bench_$v.json $out/bench_$v.txt <<'PY'
import json, sys, os
tag, j, t = sys.argv[1:4]
d = json.load(open(j))
want = os.environ.get("CLASSES", "bw1.n128,bw1.n64,wg3.n128,igemm.store.n128,other").split(",")
rows = {}
for l in open(t):
    if l.startswith('{"kernel"'):
        r = json.loads(l); rows[r["kernel"]] = r["ms_total"]
print(tag, "step", d["ms_per_step"], "median", d["step_ms"]["median"], " ".join(f"{k}={rows.get(k)}" for k in want), "serial_sum", round(sum(rows.values()), 2), flush=True)
PY
done
