O=gpurun_out/r04_c3m; mkdir -p $O
timeout -k 10 600 python3 -m pytest -x -q tests/test_timed_kernels_gpu.py -k "forward_phases or c2_c3_networks or two_block" tests/test_configs_gpu.py -k "forward_phases or c2_c3_networks or two_block or full_size_c2" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
export CLASSES=conv3.store.n64,thin.logits.n32
bash tools/sweep_lib.sh r04_c3m main 2>&1 | tee $O/sweep.txt
DMM_NO_C3_MERGE=1 bash tools/sweep_lib.sh r04_c3m_off main 2>&1 | tee -a $O/sweep.txt
bash tools/sweep_lib.sh r04_c3m2 main 2>&1 | tee -a $O/sweep.txt
DMM_NO_C3_MERGE=1 bash tools/sweep_lib.sh r04_c3m_off2 main 2>&1 | tee -a $O/sweep.txt
