#!/bin/bash
# separate processes: the process's first forward is the one that differed
out=gpurun_out/determinism.txt; : > $out
run() { timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids >> $out; }
REF32=1 run python tools/determinism.py c2 -
run python tools/determinism.py c2 thin_logits
run python tools/determinism.py c2 conv3
run python tools/determinism.py c2 cvp
run python tools/determinism.py c2 overlap_wgrad
DMM_NO_PACK_SPLIT=1 run python tools/determinism.py c2 -
run python tools/determinism.py c2 thin_logits+conv3+cvp
run python tools/determinism.py c1 -
cat $out
