python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --table --ops 2000 2>&1 >/dev/null | grep -E "cf.store|conv3.store.n32/f.b[12]" | awk '{print $1, $3, $4, $5, $6, $7}' | sort -k2 | head -30
DMM_NO_CF=1 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --table --ops 2000 2>&1 >/dev/null | grep -E "conv3.store.n32/f.b[12]" | awk '{print $1, $3, $4, $5, $6, $7}' | sort -k2 | head -30
