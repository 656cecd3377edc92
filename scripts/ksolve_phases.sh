#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { rm -rf /tmp/p_ks; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ks -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/ks.log 2>&1; python3 $R/profiles/trace_summary.py $(ls /tmp/p_ks/*/*kernel_trace.csv | head -1) | grep -E "^k_gn_solve " | head -4; }
echo "== this build"; one
for l in $R/icet_amd/lib_exp_*/libicet_hip.so; do echo "== $l"; export ICET_HIP_LIB=$l; one; done
