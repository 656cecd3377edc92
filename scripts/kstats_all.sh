#!/bin/bash
# Whole-batch per-kernel times (largest launch size of every kernel) for this build and every icet_amd/lib_exp*/ build, side by side.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { rm -rf /tmp/p_ks; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ks -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency > /tmp/ks.log 2>&1; python3 $R/profiles/trace_summary.py $(ls /tmp/p_ks/*/*kernel_trace.csv | head -1) | grep -E "^k_" | awk '!seen[$1]++ {printf "%-22s %8s %8.1f\n", $1, $2, $5}'; }
echo "== this build"; one
for l in $R/icet_amd/lib_exp*/libicet_hip.so; do echo "== $l"; export ICET_HIP_LIB=$l; one; done
