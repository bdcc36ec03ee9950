B="python3 bench.py --steps 14 --warmup 4 --no-cpu-baseline --no-profile"
run() { echo "== $1: $($B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")"; }
run stem_main
DMM_STEM_LEAF=1 run stem_leaf
run stem_main2
DMM_STEM_LEAF=1 run stem_leaf2
