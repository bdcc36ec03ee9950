#!/bin/bash
# round 5: the other BASELINE configurations at the current build (+ per-class tables)
out=gpurun_out/r05_configs; mkdir -p $out
for cfg in c3 c4 c5 c1; do
  timeout -k 10 300 python3 bench.py --config $cfg --steps 8 --warmup 3 --no-cpu-baseline --table > $out/bench_$cfg.json 2> $out/classes_$cfg.txt || exit 1
  python3 -c "import json; d=json.load(open('$out/bench_$cfg.json')); print('$cfg', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['schedule'])"
done
timeout -k 10 500 python3 -m pytest tests/test_timed_kernels_gpu.py -x -q -s -k "c2_c3_networks" > $out/nets.log 2>&1
grep -E "outside the dense|passed|failed" $out/nets.log | cut -c1-1200
