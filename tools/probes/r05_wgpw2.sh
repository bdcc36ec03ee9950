#!/bin/bash
# round 5: wgpw with the ConvTranspose phases merged: tests, then step / classes (in-tree) vs unmerged (DMM_NO_WGP_MERGE=1) vs the four-wave kernel (lab build)
out=gpurun_out/r05_wgpw2; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_timed_kernels_gpu.py -x -q -k "wave_specialised or backward_kernels_at_production or parity_phase or phases_in_one_launch or c2_c3_networks" > $out/tests.log 2>&1; rc=$?
tail -5 $out/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in new nomerge old; do
  unset DMM_LIB_PATH DMM_NO_WGPW DMM_NO_WGP_MERGE
  if [ $v = old ]; then export DMM_LIB_PATH=$PWD/build_var/lib_lab_wgp.so DMM_NO_WGPW=1; fi
  if [ $v = nomerge ]; then export DMM_NO_WGP_MERGE=1; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --table --ops 2000 > $out/bench_$v.json 2> $out/classes_$v.txt || exit 1
  python3 -c "import json; d=json.load(open('$out/bench_$v.json')); print('$v', d['ms_per_step'], d['schedule'])"
  grep -E '"kernel": "(wgp|wgrad\.n128)' $out/classes_$v.txt | cut -c1-120
  grep -E " wgp\." $out/classes_$v.txt | sort -k4 | awk '{printf "%s %s  ", $1, $3} END {print ""}'
done
