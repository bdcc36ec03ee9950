# Round 4: the fp32 "fold anomaly" bisect.  The experiment builds come from the CPU container, e.g.
#   for v in "f0:" "f1:-DFOLD_VARIANT=1" "f2:-DFOLD_VARIANT=2" "f3:-DFOLD_VARIANT=3" "f4:-mllvm -amdgpu-spill-vgpr-to-agpr=0" "f5:-fno-slp-vectorize"; do
#     bash tools/build_variant.sh fold_${v%%:*} igemm.hip:igemm_f32.o -DIGEMM_PART=0 -DIGEMM_F32_FOLD=1 ${v#*:}; done
# and by instantiation: ... -DIGEMM_FOLD_EPI=0|1 -DIGEMM_FOLD_BN=32|64|128 -DIGEMM_FOLD_LIN=0|1 (igemm.hip); compare gradients with
#   python3 tools/grad_dump.py no fp32 a.pt; DMM_LIB_PATH=build_var/lib_<variant>.so python3 tools/grad_dump.py no fp32 b.pt; python3 tools/grad_dump.py --diff a.pt b.pt
mkdir -p gpurun_out/r04_fold
for v in main f0 f1 f2 f3 f4 f5; do
  if [ "$v" = main ]; then unset DMM_LIB_PATH; else export DMM_LIB_PATH=$PWD/build_var/lib_fold_$v.so; fi
  echo "== $v" >> gpurun_out/r04_fold/dbg.txt
  timeout -k 10 120 python3 tools/probes/dbg_case.py >> gpurun_out/r04_fold/dbg.txt 2>&1
  echo "== $v" >> gpurun_out/r04_fold/tiny.txt
  timeout -k 10 200 python3 -m pytest -x -q "tests/test_model_gpu.py::test_tiny_training_step_fp32" 2>&1 | tail -4 >> gpurun_out/r04_fold/tiny.txt
done
unset DMM_LIB_PATH
timeout -k 10 300 python3 tools/probes/thin_flaky.py > gpurun_out/r04_fold/thin_flaky.txt 2>&1
cat gpurun_out/r04_fold/dbg.txt | grep -v amdgpu.ids | cut -c1-200; cat gpurun_out/r04_fold/tiny.txt | cut -c1-200; cat gpurun_out/r04_fold/thin_flaky.txt
