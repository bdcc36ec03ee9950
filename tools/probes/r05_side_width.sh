#!/bin/bash
# round 5: how wide should the side stream's persistent kernels be?  One workgroup per CU (the defaults) blocks the main chain's small
# launches for the whole launch; narrower launches are slower alone but leave CUs to the chain.  Lab build; 12 steps each.
out=gpurun_out/r05_side_width; mkdir -p $out
export DMM_LIB_PATH=$PWD/build_var/lib_lab.so
run() { tag=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-profile > $out/b_$tag.json 2> $out/b_$tag.err || exit 1
  python3 -c "import json; d=json.load(open('$out/b_$tag.json')); print('$tag', d['ms_per_step'], d['step_ms'])"; }
run base X=1
run wgp128 DMM_WGP_WGS=128
run wgp160 DMM_WGP_WGS=160
run wg3_128 DMM_WG3_WGS=128
run wg3_192 DMM_WG3_WGS=192
run wg5_1 DMM_WG5_PER_CU=1
run all128 DMM_WGP_WGS=128 DMM_WG3_WGS=128 DMM_WG5_PER_CU=1
run all160 DMM_WGP_WGS=160 DMM_WG3_WGS=160 DMM_WG5_PER_CU=1
run base2 X=1
