# kernel trace of a few steps + the timeline summary ("What a step is made of"):  bash tools/probes/r04_trace.sh [config ...]
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r04_trace; mkdir -p $O; cd /tmp
for c in ${@:-c2}; do
  rm -rf /tmp/tr_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o run -- python3 $R/bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/trace_$c.log 2>&1 || exit 1
  python3 $R/tools/trace_timeline.py $(find /tmp/tr_$c -name "*kernel_trace.csv" | head -1) > $O/${c}_timeline.txt || exit 2
  head -4 $O/${c}_timeline.txt
done
python3 $R/tools/host_bound.py c2 c5 c4 c2@64x96 c5@64x96 c4@64x96 2>&1 | grep -v amdgpu.ids | tee $O/host_bound.txt
