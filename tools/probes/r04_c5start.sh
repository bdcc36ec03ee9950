# the first / last kernels of a step per hardware queue (what the queues wait for at the two ends):  bash tools/probes/r04_c5start.sh [config]
export TMPDIR=/tmp; R=$PWD; C=${1:-c5}; O=$R/gpurun_out/r04_trace; mkdir -p $O; cd /tmp; rm -rf /tmp/tr5
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr5 -o run -- python3 $R/bench.py --config $C --steps 4 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2>&1
T=$(find /tmp/tr5 -name "*kernel_trace.csv" | head -1)
python3 $R/tools/probes/dump_first.py $T 400 > $O/${C}_first.txt
python3 $R/tools/probes/dump_first.py $T -60 > $O/${C}_last.txt
python3 $R/tools/trace_timeline.py $T > $O/${C}_timeline.txt
head -30 $O/${C}_timeline.txt
