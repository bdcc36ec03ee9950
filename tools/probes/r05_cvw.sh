#!/bin/bash
# round 5: cvw.hip bring-up - short timeouts (a barrier mismatch would hang the workgroup): smallest cases first
out=gpurun_out/r05_cvw; mkdir -p $out
timeout -k 5 120 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "parity_phase_weight_gradients_on_lds_tiles" > $out/t1.log 2>&1; rc=$?
tail -15 $out/t1.log | cut -c1-400
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 5 300 python3 -m pytest tests/test_timed_kernels_gpu.py -x -q -k "forward_kernels_at_production or decoder_transposed_convolution_phases or two_block or deferred_head" > $out/t2.log 2>&1; rc=$?
tail -8 $out/t2.log | cut -c1-400
if [ $rc -ne 0 ]; then exit $rc; fi
for v in new old; do
  unset DMM_LIB_PATH DMM_NO_CVW
  if [ $v = old ]; then export DMM_LIB_PATH=$PWD/build_var/lib_lab_cvp.so DMM_NO_CVW=1; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --table --ops 2000 > $out/bench_$v.json 2> $out/classes_$v.txt || exit 1
  python3 -c "import json; d=json.load(open('$out/bench_$v.json')); print('$v', d['ms_per_step'], d['schedule']['serial_kernel_sum_ms'])"
  grep -E " cvp.store" $out/classes_$v.txt | head -5
done
