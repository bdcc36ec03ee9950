# kernel trace of a few steps + the timeline summary ("What a step is made of")
O=gpurun_out/r04_trace; mkdir -p $O; export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o run -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $R/$O/trace.log 2>&1
python3 $R/tools/trace_timeline.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) | tee $R/$O/timeline.txt
