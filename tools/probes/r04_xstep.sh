B="python3 bench.py --steps 14 --warmup 4 --no-cpu-baseline --no-profile"
run() { echo "== $1: $($B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")"; }
run base
DMM_LIB_PATH=$PWD/build_var/lib_xstep.so run xstep
run base2
DMM_LIB_PATH=$PWD/build_var/lib_xstep.so run xstep2
for c in c5 c3; do
echo "$c base: $(python3 bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
echo "$c xstep: $(DMM_LIB_PATH=$PWD/build_var/lib_xstep.so python3 bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
done
