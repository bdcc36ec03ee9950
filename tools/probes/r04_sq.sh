export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/pmcS
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d /tmp/pmcS -o run -- $B > /dev/null 2>&1 || exit 6
cd $R; python3 tools/pmc_sq.py /tmp/pmcS gpurun_out/r04_sq.csv || exit 7
head -1 gpurun_out/r04_sq.csv; grep -E "hf_kernel|cf_kernel|wg3_kernel|conv3_kernelIDF16_Li16ELi2ELin1ELi1ELi1ELi4ELi0" gpurun_out/r04_sq.csv
rm -rf /tmp/pmcL
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/pmcL -o run -- $B > /dev/null 2>&1 || exit 7
python3 tools/pmc_lds.py /tmp/pmcL | grep -E "kernel|hf_kernel|cf_kernel|wg3_kernel" | head
