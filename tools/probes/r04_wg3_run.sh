mkdir -p gpurun_out/r04_wg3; O=gpurun_out/r04_wg3
timeout -k 10 400 python3 -m pytest -x -q tests/test_timed_kernels_gpu.py -k "dense_3x3_weight or backward_kernels_at_production or c2_c3_networks" > $O/pytest_wg3.log 2>&1; tail -5 $O/pytest_wg3.log
timeout -k 10 150 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --table --ops 2000 > $O/bench_slots.json 2> $O/bench_slots_ops.txt || exit 3
DMM_WG3_SLOTS=0 timeout -k 10 150 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --table > $O/bench_atomics.json 2> $O/bench_atomics_ops.txt || exit 4
for f in slots atomics; do python3 - $O/bench_$f.json $O/bench_${f}_ops.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rows={}
for l in open(sys.argv[2]):
    if l.startswith('{"kernel"'):
        r=json.loads(l); rows[r["kernel"]]=r["ms_total"]
print(sys.argv[1], d["ms_per_step"], "wg3", rows.get("wg3.n128"), "serial_sum", round(sum(rows.values()),2))
PY
done
for v in f0 f3; do echo "== $v" >> $O/lab_fold.txt; DMM_LIB_PATH=$PWD/build_var/lib_fold_$v.so timeout -k 10 300 python3 tools/gpu_lab.py kernels 2>&1 | grep -v "^ok " >> $O/lab_fold.txt; done
DMM_LIB_PATH=$PWD/build_var/lib_fold_f3.so timeout -k 10 100 python3 -m pytest -x -q "tests/test_model_gpu.py::test_tiny_training_step_fp32" 2>&1 | grep -E "^E  |Error" | head -8 >> $O/lab_fold.txt
cat $O/lab_fold.txt | cut -c1-300
