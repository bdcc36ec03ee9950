O=gpurun_out/r04_2p; mkdir -p $O
timeout -k 10 600 python3 -m pytest -x -q tests/test_timed_kernels_gpu.py -k "two_pass or c2_c3_networks or two_block" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
export CLASSES=conv3.bnbwd.n64,applycorr,wg3.n128,other
bash tools/sweep_lib.sh r04_2p main 2>&1 | tee $O/sweep.txt
DMM_NO_TWO_PASS=1 bash tools/sweep_lib.sh r04_2p_off main 2>&1 | tee -a $O/sweep.txt
DMM_WG3_WGS=128 bash tools/sweep_lib.sh r04_2p_w128 main 2>&1 | tee -a $O/sweep.txt
DMM_NO_OVERLAP=1 bash tools/sweep_lib.sh r04_2p_noov main 2>&1 | tee -a $O/sweep.txt
