#!/bin/bash
out=gpurun_out/r05_rawstats; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_timed_kernels_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py -x -q -k "raw_input_batchnorm or eight_channel or c2_c3_networks or two_block or d121 or c3_network or c5 or golden or tiny_training" > $out/tests.log 2>&1; rc=$?
tail -6 $out/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in new old; do
  unset DMM_NO_RAW_STATS
  if [ $v = old ]; then export DMM_NO_RAW_STATS=1; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --table --ops 2000 > $out/bench_$v.json 2> $out/classes_$v.txt || exit 1
  python3 -c "import json; d=json.load(open('$out/bench_$v.json')); print('$v', d['ms_per_step'], d['schedule']['serial_kernel_sum_ms'])"
  grep -E "wg5|bnbwd.n32/h" $out/classes_$v.txt | grep -v kernel | head
done
