#!/bin/bash
# Instruction-cache counters of the default bench's kernels (one --pmc pass).  Run through gpurun: bash scripts/pmc_icache.sh > gpurun_out/pmc_icache.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0"
rm -rf /tmp/p_ic
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/p_ic -- $CMD > /tmp/ic.log 2>&1 || { echo "pass failed"; tail -3 /tmp/ic.log; exit 1; }
f=$(ls /tmp/p_ic/*/*counter_collection.csv | head -1)
grep -E "Kernel_Name|icet::" $f > /tmp/p_ic.csv
python3 $R/profiles/pmc_summary.py /tmp/p_ic.csv
