O=gpurun_out/r04_c5; mkdir -p $O
for s in 1 0; do
  DMM_WG3_SLOTS=$s timeout -k 10 200 python3 bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --table --ops 3000 > $O/bench_s$s.json 2> $O/bench_s$s.txt
  python3 - $O/bench_s$s.json $O/bench_s$s.txt <<'PY'
import json,sys,re,collections
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
blk=collections.defaultdict(lambda:[0,0.0])
for l in open(sys.argv[2]):
    m=re.match(r'\s+([\d.]+) ms\s+wg3\.n128/(f|s2)\.b(\d)\.', l)
    if m: blk[m.group(3)][0]+=1; blk[m.group(3)][1]+=float(m.group(1))
print(sys.argv[1], d['ms_per_step'], {k:(n,round(1000*t/n,1)) for k,(n,t) in sorted(blk.items())})
PY
done
