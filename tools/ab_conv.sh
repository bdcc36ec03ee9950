#!/bin/bash
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  if [ "$v" = main ]; then unset DMM_LIB_PATH; else export DMM_LIB_PATH=$PWD/build_var/lib_$v.so; fi
  echo "== $v"; DMM_FAT_WGS=0 timeout -k 10 120 python3 tools/conv_time.py 20 2>&1 | grep -v amdgpu.ids
done > $out/conv_time.txt 2>&1
