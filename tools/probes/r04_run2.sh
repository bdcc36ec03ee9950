O=gpurun_out/r04_wg3b; mkdir -p $O
export CLASSES=wg3.n128,bw1.reduce,other
bash tools/sweep_lib.sh r04_wg3b main wg3d1 wg3d2 wg3d4 wg3d6 wg3d7 2>&1 | tee $O/sweep.txt
DMM_WG3_SLOTS=0 bash tools/sweep_lib.sh r04_wg3b_atom main wg3d1 2>&1 | tee $O/sweep_atomics.txt
grep "wg3.n128/f.b[1234].l2.conv2" gpurun_out/r04_wg3b/bench_main.txt gpurun_out/r04_wg3b/bench_wg3d*.txt | cut -c1-200 > $O/per_block.txt; cat $O/per_block.txt
python3 tools/grad_dump.py no fp32 /tmp/a.pt && DMM_LIB_PATH=$PWD/build_var/lib_fold_f0.so python3 tools/grad_dump.py no fp32 /tmp/b.pt && python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/b.pt > $O/fold_diff.txt 2>&1; cat $O/fold_diff.txt | head -80
