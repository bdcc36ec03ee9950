#!/bin/bash
# A/B of an environment knob on the GPU box: tools/ab_env.sh <out-tag> <VAR> <value...>
tag=$1; var=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out
for v in "$@"; do
  export $var=$v
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline ${AB_ARGS} > $out/bench_$v.json 2> $out/bench_$v.txt || exit 1
  python3 -c "import json,sys; d=json.load(open('$out/bench_$v.json')); print('$var=$v', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('frac_alone'), d.get('encoder_1x1'))"
done
