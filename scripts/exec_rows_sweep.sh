#!/bin/bash
# k_exec_flags_pair with 4 / 6 / 8 (shipped) / 12 / 16 rows per thread (icet_amd/lib_exp_r*/, make EXTRA=-DICET_EXEC_PAIR_ROWS=N): its time per launch and the step (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { rm -rf /tmp/p_ex; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ex -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-h2d --traffic none --min-timed-s 0 > /tmp/ex.log 2>&1; python3 $R/profiles/trace_summary.py $(ls /tmp/p_ex/*/*kernel_trace.csv | head -1) | grep -E "^k_exec_flags_pair|^k_scramble_src " | head -2; grep "^{\"metric\"" /tmp/ex.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   step', d['ms_per_step'])"; }
echo "== shipped (8 rows)"; one
for l in $R/icet_amd/lib_exp_r*/libicet_hip.so; do echo "== $l"; export ICET_HIP_LIB=$l; one; done
