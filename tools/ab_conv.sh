#!/bin/bash
# tools/ab_conv.sh <out-tag> <VAR> <value...>: tools/conv_time.py under each value of an environment knob ("-" = unset)
out=gpurun_out/$1; var=$2; shift 2; mkdir -p $out
for v in "$@"; do
  if [ "$v" = - ]; then unset $var; else export $var=$v; fi
  echo "== $var=$v"; timeout -k 10 120 python3 tools/conv_time.py 20 2>&1 | grep -v amdgpu.ids
done > $out/conv_time.txt 2>&1
