#!/bin/bash
# Per-kernel whole-batch times of one kernel (argument 1, e.g. k_gn_solve) for this build and every icet_amd/lib_exp*/ build.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { rm -rf /tmp/p_ks; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ks -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency > /tmp/ks.log 2>&1; python3 $R/profiles/trace_summary.py $(ls /tmp/p_ks/*/*kernel_trace.csv | head -1) | grep -E "^$1 " | head -3; }
echo "== this build"; one "$1"
for l in $R/icet_amd/lib_exp*/libicet_hip.so; do echo "== $l"; export ICET_HIP_LIB=$l; one "$1"; done
