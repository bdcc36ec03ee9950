#!/bin/bash
# round 5: what the head's 5x5 data gradient passes (conv3.hip thin-only variant, 76 800 one-tile workgroups) spend their time on
out=gpurun_out/r05_c3dbg; mkdir -p $out
for v in main c3dbg4 c3dbg2 c3dbg6 c3dbg8 c3dbg16; do
  if [ $v = main ]; then unset DMM_LIB_PATH; else export DMM_LIB_PATH=$PWD/build_var/lib_$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --table --ops 2000 > $out/bench_$v.json 2> $out/classes_$v.txt || exit 1
  echo "== $v"; grep -E "conv3.bnbwd.n64/h.refine1|conv3.bnbwd.n128/f.b1.l3|conv3.store.n64/f.conv0" $out/classes_$v.txt | head -4
done
