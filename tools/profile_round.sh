#!/bin/bash
# Runs on the GPU box: bench + rocprofv3 kernel-trace stats + PMC passes; leaves only small summaries under gpurun_out/$1.
set -o pipefail
tag=${1:-r01}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile"
timeout -k 10 300 python3 bench.py --table --ops 2000 > $out/bench.json 2> $out/bench_classes_and_ops.txt || exit 1
cat $out/bench.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- $B > $out/prof.log 2>&1 || exit 2
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 - $out/kernel_stats.csv $out/kernel_stats.json $out/prof.log <<'PY' || exit 2
# the in-step kernel-time sum bench.py reports under "schedule": every dmm kernel of all steps (timed + warm-up) of the profiled command.
# Workload, batch, dtype and step count come from the JSON line that very command printed; the source stamp ties the file to this build.
import csv, json, sys
sys.path.insert(0, ".")
from tools.src_hash import source_hash
line = json.loads([l for l in open(sys.argv[3]) if l.startswith("{") and '"metric"' in l][-1])
steps = line["steps"] + line["warmup"]
cfg = line["config"]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dmm" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in rows) / 1e6
json.dump({"config": cfg["key"], "batch": cfg["per_gpu_batch"], "dtype": line["dtype"], "steps": steps, "total_kernel_ms": round(tot, 3),
           "source_sha16": source_hash(), "workload": cfg["workload"],
           "command": "rocprofv3 --kernel-trace --stats -- " + " ".join(sys.argv[4:]) if len(sys.argv) > 4 else "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile",
           "note": "all dmm:: kernels of both streams, %d steps" % steps}, open(sys.argv[2], "w"), indent=1)
print("in-step kernel sum per step: %.2f ms" % (tot / steps))
PY
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -o run -- $B > $out/pmcF.log 2>&1 || exit 3
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -o run -- $B > $out/pmcW.log 2>&1 || exit 4
python3 tools/pmc_traffic.py /tmp/pmcF /tmp/pmcW $out/pmc_hbm_traffic.json > $out/pmc_hbm_traffic.txt || exit 5
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d /tmp/pmcS -o run -- $B > $out/pmcS.log 2>&1 || exit 6
python3 tools/pmc_sq.py /tmp/pmcS $out/pmc_sq_summary.csv || exit 7
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/pmcL -o run -- $B > $out/pmcL.log 2>&1 || exit 7
python3 tools/pmc_lds.py /tmp/pmcL > $out/pmc_lds_bank_conflicts.txt || exit 7
for cfg in c1 c3 c4 c5; do
  timeout -k 10 240 python3 bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --table > $out/bench_$cfg.json 2> $out/bench_${cfg}_classes.txt || exit 8
done
DMM_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $out/bench_c2_forced_dist_world1.json 2> $out/bench_c2_forced_dist.log || exit 9
ls -la $out
