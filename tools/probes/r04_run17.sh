O=gpurun_out/r04_defer; mkdir -p $O
timeout -k 10 600 python3 -m pytest -x -q tests/test_timed_kernels_gpu.py -k "deferred or two_pass or c2_c3_networks or two_block" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-profile"
run() { echo "== $1: $($B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")"; }
run default
DMM_DEFER_WGRAD=0 run nodefer
DMM_DEFER_AT=features.transition3 run at_transition3
DMM_DEFER_AT=features.denseblock3.denselayer24 run at_block3
DMM_DEFER_AT=features.denseblock2.denselayer12 run at_block2
DMM_DEFER_AT=features.denseblock4.denselayer8. run at_block4_mid
