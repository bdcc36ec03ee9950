O=gpurun_out/r04_tests; mkdir -p $O
timeout -k 10 900 python3 -m pytest -x -q tests/test_timed_kernels_gpu.py tests/test_next_rows.py > $O/pytest.log 2>&1; tail -15 $O/pytest.log | cut -c1-300
grep -h "FAIL" $O/pytest.log | head -20 | cut -c1-300
