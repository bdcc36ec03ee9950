# A/B of the pack cut / encoder interleave (round 4): same box, alternating
mkdir -p gpurun_out/r04_packcut; OUT=gpurun_out/r04_packcut/ab.txt; : > $OUT
run() { # name config
  v=$(python3 bench.py --config $2 --steps 14 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")
  echo "== $2 $1: $v" | tee -a $OUT
}
timeout -k 10 300 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT
for c in c5 c4 c3 c2 c1; do
  run new $c
  DMM_PACK_CUT=1 DMM_NO_S2_INTERLEAVE=1 run old $c
  DMM_NO_S2_INTERLEAVE=1 run cut_only $c
  DMM_PACK_CUT=1 run interleave_only $c
  run new2 $c
done
