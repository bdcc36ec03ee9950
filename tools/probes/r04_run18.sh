O=gpurun_out/r04_wgpm; mkdir -p $O
timeout -k 10 600 python3 -m pytest -x -q tests/test_timed_kernels_gpu.py -k "one_launch or parity_phase or c2_c3_networks" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
export CLASSES=wgp.n64,wgp.n128,wg5.n64
bash tools/sweep_lib.sh r04_wgpm main 2>&1 | tee $O/sweep.txt
DMM_NO_WGP_MERGE=1 bash tools/sweep_lib.sh r04_wgpm_off main 2>&1 | tee -a $O/sweep.txt
grep "wgp.n64" gpurun_out/r04_wgpm/bench_main.txt gpurun_out/r04_wgpm_off/bench_main.txt | cut -c1-150
