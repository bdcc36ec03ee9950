B="python3 bench.py --steps 14 --warmup 4 --no-cpu-baseline --no-profile"
run() { echo "== $1: $($B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")"; }
run base
DMM_WGP_WGS=128 run wgp128
DMM_WGP_WGS=192 run wgp192
DMM_WGRAD_WGS=128 run wgrad128
DMM_WGRAD_WGS=384 run wgrad384
DMM_WG3_WGS=128 run wg3_128
DMM_WG3_WGS=192 run wg3_192
DMM_WG3_SLOTS=0 run wg3_atomics
DMM_BW1_PER_CU=1 run bw1_1
DMM_BW1_PER_CU=3 run bw1_3
DMM_C3_ROUNDS=2 run c3r2
DMM_PIG_PER_CU=1 run pig1
run base2
