#!/bin/bash
# A/B of experiment builds on the GPU box: tools/ab.sh <out-tag> <lib...>; "main" = the in-tree library
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
for v in "$@"; do
  if [ "$v" = main ]; then unset DMM_LIB_PATH; else export DMM_LIB_PATH=$PWD/build_var/lib_$v.so; fi
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline ${AB_ARGS} > $out/bench_$v.json 2> $out/bench_$v.txt || exit 1
  python3 -c "import json,sys; d=json.load(open('$out/bench_$v.json')); print('$v', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('frac_alone'), d.get('encoder_1x1'))"
done
