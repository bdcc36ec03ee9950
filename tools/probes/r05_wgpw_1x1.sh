#!/bin/bash
out=gpurun_out/r05_wgpw_1x1; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_timed_kernels_gpu.py tests/test_configs_gpu.py -x -q -k "c2_c3_networks or two_block or d121 or mid_fusion_networks" > $out/tests.log 2>&1; rc=$?
tail -4 $out/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in new old; do
  unset DMM_LIB_PATH DMM_NO_WGPW_1X1
  if [ $v = old ]; then export DMM_LIB_PATH=$PWD/build_var/lib_lab_wgp.so DMM_NO_WGPW_1X1=1; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --table --ops 2000 > $out/bench_$v.json 2> $out/classes_$v.txt || exit 1
  python3 -c "import json; d=json.load(open('$out/bench_$v.json')); print('$v', d['ms_per_step'], d['schedule']['serial_kernel_sum_ms'])"
  grep -E "conv_reduce" $out/classes_$v.txt | grep -E "wgp|wgrad" | head
done
