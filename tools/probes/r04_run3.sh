O=gpurun_out/r04_wg3c; mkdir -p $O
timeout -k 10 400 python3 -m pytest -x -q tests/test_timed_kernels_gpu.py -k "dense_3x3_weight or backward_kernels_at_production" > $O/pytest_wg3.log 2>&1; tail -3 $O/pytest_wg3.log
export CLASSES=wg3.n128,bw1.reduce,other
bash tools/sweep_lib.sh r04_wg3c main 2>&1 | tee $O/sweep.txt
grep "wg3.n128/f.b[1234].l2.conv2" gpurun_out/r04_wg3c/bench_main.txt | cut -c1-120
python3 tools/grad_dump.py no fp32 /tmp/a.pt && DMM_LIB_PATH=$PWD/build_var/lib_fold_f0.so python3 tools/grad_dump.py no fp32 /tmp/b.pt && python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/b.pt > $O/fold_diff.txt 2>&1; head -60 $O/fold_diff.txt
