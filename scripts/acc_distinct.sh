#!/bin/bash
# k_gn_accumulate against the number of DISTINCT pairs in the 256-pair batch (1 = every block of a round does the same work, scans cache-resident).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo -n "$* : "; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0.1 $* 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'acc_ms', r['avg_launch_ms'], 'frac', r['frac'], 'kf', r['keyframe_ms_per_step'], 'gn', r['gn_loop_ms_per_step'])"; }
run --distinct 0
for v in "$@"; do run $v; done
