# hf.hip: tests, then version 2 (weights in registers) against version 1 (DMM_HF_V=1) and conv3.hip (DMM_NO_HF=1), same box
mkdir -p gpurun_out/r04_hf; OUT=gpurun_out/r04_hf/ab.txt; : > $OUT
timeout -k 10 400 python3 -m pytest tests/test_timed_kernels_gpu.py -x -q -s -m gpu -k "wave_specialised or c2_c3_networks" > gpurun_out/r04_hf/test.log 2>&1; echo "test rc=$? aperture=$(grep -c APERTURE gpurun_out/r04_hf/test.log)" | tee -a $OUT; grep -E "hf.hip vs|passed|failed" gpurun_out/r04_hf/test.log | tee -a $OUT
grep -q " passed" gpurun_out/r04_hf/test.log || exit 1
run() { v=$(python3 bench.py --config $2 --steps 14 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])"); echo "== $2 $1: $v" | tee -a $OUT; }
python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --table --ops 2000 2>&1 >/dev/null | grep -E 'h.refine0' | grep -E "store" | sed "s/^/v2 /" | tee -a $OUT

for c in c2 c5; do run v2 $c; DMM_NO_HF=1 run conv3 $c; run v2b $c; done
