#!/bin/bash
# Loop kernels of the small-batch launches of the default bench (latency / highres / real-pair records) with and without option fuse_solve.  Run through gpurun.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for f in 1 0; do
  echo "== fuse_solve=$f"
  rm -rf /tmp/p_ks; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ks -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --set fuse_solve=$f > /tmp/ks.log 2>&1
  python3 $R/profiles/trace_summary.py $(ls /tmp/p_ks/*/*kernel_trace.csv | head -1) | grep -E "^k_gn_|^k_init" | head -14
  tail -1 /tmp/ks.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('latency ms', d['latency']['ms_per_pair'], 'loop', d['latency']['gn_loop_ms'], 'highres', d['highres']['ms_per_pair'])"
done
