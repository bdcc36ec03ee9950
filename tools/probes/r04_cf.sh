mkdir -p gpurun_out/r04_cf; OUT=gpurun_out/r04_cf/ab.txt; : > $OUT
timeout -k 10 500 python3 -m pytest tests/test_timed_kernels_gpu.py -x -q -s -m gpu -k "forward_kernels_at_production_size" > gpurun_out/r04_cf/test.log 2>&1; echo "test rc=$? aperture=$(grep -c APERTURE gpurun_out/r04_cf/test.log)" | tee -a $OUT; grep -E "passed|failed|FAIL" gpurun_out/r04_cf/test.log | tail -5 | tee -a $OUT
grep -q "passed" gpurun_out/r04_cf/test.log || exit 1
run() { v=$(python3 bench.py --config $2 --steps 14 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])"); echo "== $2 $1: $v" | tee -a $OUT; }
for v in 0 1; do if [ $v = 1 ]; then export DMM_NO_CF=1; else unset DMM_NO_CF; fi
python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --table --ops 2000 2>&1 >/dev/null | grep -E '^\{"kernel": "(cf.store|conv3.store)' | cut -c1-100 | sed "s/^/nocf=$v /" | tee -a $OUT; done
unset DMM_NO_CF
for c in c2 c5; do run cf $c; DMM_NO_CF=1 run conv3 $c; run cf2 $c; DMM_NO_CF=1 run conv3b $c; done
