mkdir -p gpurun_out/r04_hf; OUT=gpurun_out/r04_hf/ab.txt; : > $OUT
run() { v=$(python3 bench.py --config $2 --steps 14 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])"); echo "== $2 $1: $v" | tee -a $OUT; }
for v in 0 1; do if [ $v = 1 ]; then export DMM_NO_HF=1; else unset DMM_NO_HF; fi
python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --table --ops 2000 2>&1 >/dev/null | grep -E 'h.refine0|f.conv0' | grep -E "store" | sed "s/^/nohf=$v /" | tee -a $OUT; done
unset DMM_NO_HF
for c in c2 c4 c5; do run hf $c; DMM_NO_HF=1 run conv3 $c; run hf2 $c; DMM_NO_HF=1 run conv3b $c; done
