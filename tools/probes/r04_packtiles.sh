# tile kernels for pack / unpack: parity tests, then A/B per config on one box
mkdir -p gpurun_out/r04_packtiles; OUT=gpurun_out/r04_packtiles/ab.txt; : > $OUT
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py tests/test_timed_kernels_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee -a $OUT
run() { # name config
  v=$(python3 bench.py --config $2 --steps 14 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")
  echo "== $2 $1: $v" | tee -a $OUT
}
for c in c2 c5 c4 c3; do
  run tiles $c
  DMM_NO_PACK_TILES=1 run generic $c
  run tiles2 $c
  DMM_NO_PACK_TILES=1 run generic2 $c
done
