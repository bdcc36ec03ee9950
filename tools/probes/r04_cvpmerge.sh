mkdir -p gpurun_out/r04_cvpmerge; OUT=gpurun_out/r04_cvpmerge/ab.txt; : > $OUT
timeout -k 10 600 python3 -m pytest tests/test_timed_kernels_gpu.py -x -q -m gpu -k "one_launch or c2_c3_networks or production" 2>&1 | tail -4 | tee -a $OUT
run() { v=$(python3 bench.py --config $2 --steps 14 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])"); echo "== $2 $1: $v" | tee -a $OUT; }
for c in c2 c5 c4; do run merged $c; DMM_NO_CVP_MERGE=1 run separate $c; run merged2 $c; DMM_NO_CVP_MERGE=1 run separate2 $c; done
python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --table --ops 2000 2>&1 >/dev/null | grep -E "cvp.store" | tee -a $OUT
