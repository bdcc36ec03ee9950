O=gpurun_out/r04_mid; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for c in c2 c4 c5; do timeout -k 10 150 python3 bench.py --config $c --steps 12 --warmup 4 --no-cpu-baseline --table > $O/bench_$c.json 2> $O/bench_${c}_classes.txt || exit 8; python3 -c "
import json
d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'], d['step_ms'])"; done
