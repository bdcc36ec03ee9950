#!/bin/bash
out=gpurun_out/r05_cwdbg; mkdir -p $out
for v in main cwdbg1 cwdbg2 cwdbg4 cwdbg6 cwdbg8 cwdbg16; do
  if [ $v = main ]; then unset DMM_LIB_PATH; else export DMM_LIB_PATH=$PWD/build_var/lib_$v.so; fi
  timeout -k 10 120 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --table --ops 2000 > $out/bench_$v.json 2> $out/classes_$v.txt || exit 1
  echo "== $v $(grep -E ' cvp.store' $out/classes_$v.txt | sort -k3 | awk '{printf "%s %s  ", $3, $1}')"
done
