for c in c5 c2; do for v in 0 1; do
if [ $v = 1 ]; then export DMM_NO_PACK_TILES=1; else unset DMM_NO_PACK_TILES; fi
python3 bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --table --ops 2000 2>&1 >/dev/null | grep -E '^\{"kernel": "(pack|unpack)"' | sed "s/^/$c notiles=$v /"
done; done
