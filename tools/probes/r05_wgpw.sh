#!/bin/bash
# round 5: wgpw.hip bring-up on the GPU box: parity tests, then the step and the wgp classes with the wave-specialised form (in-tree
# library) against the four-wave kernel (lab build of wgp.hip with DMM_NO_WGPW=1)
out=gpurun_out/r05_wgpw; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_timed_kernels_gpu.py -x -q -k "wave_specialised or backward_kernels_at_production or parity_phase or phases_in_one_launch" > $out/tests.log 2>&1; rc=$?
tail -5 $out/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in new old; do
  if [ $v = old ]; then export DMM_LIB_PATH=$PWD/build_var/lib_lab_wgp.so DMM_NO_WGPW=1; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --table > $out/bench_$v.json 2> $out/classes_$v.txt || exit 1
  python3 -c "import json; d=json.load(open('$out/bench_$v.json')); print('$v', d['ms_per_step'], d['schedule'])"
  grep -E "wgp|wgrad\.n128" $out/classes_$v.txt | head -8
done
