for m in "" h e a; do
  if [ -z "$m" ]; then unset DMM_SKIP_WGRAD; else export DMM_SKIP_WGRAD=$m; fi
  echo "== skip '$m' $(timeout -k 10 120 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'])")"
done
