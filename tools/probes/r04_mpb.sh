timeout -k 10 400 python3 -m pytest tests/test_model_gpu.py tests/test_timed_kernels_gpu.py -x -q -m gpu -k "not production" 2>&1 | tail -2
python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --table --ops 2000 2>&1 >/dev/null | grep -E '^\{"kernel": "(maxpool.bwd|maxpool.fwd)"' | cut -c1-110
for i in 1 2; do python3 bench.py --steps 14 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])"; done
