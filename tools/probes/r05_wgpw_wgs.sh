#!/bin/bash
# round 5: workgroups per wgp launch with the wave-specialised kernel (a tile is ~3x shorter, the 131-262 KB of end-of-walk atomics per workgroup are not)
out=gpurun_out/r05_wgpw_wgs; mkdir -p $out
export DMM_LIB_PATH=$PWD/build_var/lib_lab_wgp.so
for v in 48 64 96 128 192 256; do
  export DMM_WGP_WGS=$v
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --table --ops 2000 > $out/bench_$v.json 2> $out/classes_$v.txt || exit 1
  python3 -c "import json; d=json.load(open('$out/bench_$v.json')); print('$v', d['ms_per_step'], d['schedule']['serial_kernel_sum_ms'])"
  grep -E '"kernel": "wgp' $out/classes_$v.txt | cut -c1-110
  grep -E " wgp\." $out/classes_$v.txt | sort -k4 | awk '{printf "%s %s  ", $1, $3} END {print ""}'
done
