O=gpurun_out/r04_eff; mkdir -p $O
timeout -k 10 600 python3 -m pytest -x -q tests/test_timed_kernels_gpu.py -k "compact or dense_3x3 or backward_kernels_at_production or c2_c3_networks or two_block" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
export CLASSES=wg3.n128,conv3.bnbwd.n128,bw1.reduce,other
bash tools/sweep_lib.sh r04_eff main 2>&1 | tee $O/sweep.txt
DMM_NO_EFF_COMPACT=1 bash tools/sweep_lib.sh r04_eff_off main 2>&1 | tee -a $O/sweep.txt
grep "wg3.n128/f.b[1234].l2.conv2\|conv3.bnbwd.n128/f.b[1234].l2.conv2" gpurun_out/r04_eff/bench_main.txt gpurun_out/r04_eff_off/bench_main.txt | cut -c1-140
python3 tools/grad_dump.py no fp32 /tmp/a.pt && DMM_LIB_PATH=$PWD/build_var/lib_fold_f0.so python3 tools/grad_dump.py no fp32 /tmp/b.pt && python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/b.pt > $O/fold_diff.txt 2>&1; head -70 $O/fold_diff.txt
