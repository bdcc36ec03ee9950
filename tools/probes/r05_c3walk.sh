#!/bin/bash
# round 5: conv3.hip data-gradient variants walk tiles (one atomic round per workgroup) - tests, then classes against the previous build
out=gpurun_out/r05_c3walk; mkdir -p $out
timeout -k 10 800 python3 -m pytest tests/test_timed_kernels_gpu.py tests/test_kernels_gpu.py -x -q -k "production or dense_3x3 or two_pass or eight_channel or c2_c3_networks or two_block or kernels" > $out/tests.log 2>&1; rc=$?
tail -4 $out/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --table --ops 2000 > $out/bench_new.json 2> $out/classes_new.txt || exit 1
python3 -c "import json; d=json.load(open('$out/bench_new.json')); print('new', d['ms_per_step'], d['schedule'])"
grep -E '"kernel": "conv3' $out/classes_new.txt | cut -c1-130
grep -E "conv3\.(bnbwd|store)\.n(64|32)/(h\.|f.conv0)" $out/classes_new.txt | head
