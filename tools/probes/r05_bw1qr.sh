#!/bin/bash
# round 5 (record of an experiment; profiles/r05/ablations.txt): bw1 computing norm2's correction constants itself (the finalize launch off the
# chain) against the tables, same box.  The switch DMM_NO_BW1_QR existed only in the experiment build (reverted: slower).
out=gpurun_out/r05_bw1qr; mkdir -p $out
for rep in 1 2; do
for v in new old; do
  unset DMM_NO_BW1_QR; if [ $v = old ]; then export DMM_NO_BW1_QR=1; fi
  for cfg in c2 c5; do
    timeout -k 10 200 python3 bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-profile > $out/${cfg}_${v}_$rep.json 2> $out/${cfg}_${v}_$rep.err || exit 1
    python3 -c "import json; d=json.load(open('$out/${cfg}_${v}_$rep.json')); print('$cfg $v $rep', d['ms_per_step'], d['step_ms']['median'])"
  done
done
done
