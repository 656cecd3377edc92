#!/bin/bash
# Per-kernel times of the default bench under rocprofv3 (whole-batch launches listed per launch size).  Run through gpurun.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p_ks; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ks -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency "$@" > /tmp/ks.log 2>&1
python3 $R/profiles/trace_summary.py $(ls /tmp/p_ks/*/*kernel_trace.csv | head -1) | head -70
