#!/bin/bash
# k_gn_accumulate launch-shape sweep (timing mode).  usage through gpurun: acc_sweep.sh "lds_slots=600 acc_blocks=1024" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo -n "$* : "; a=""; for kv in $*; do a="$a --set $kv"; done; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0.1 $a 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'acc_ms', r['avg_launch_ms'], 'frac', r['frac'], 'gn', r['gn_loop_ms_per_step'])"; }
run "acc_pts=4"
for v in "$@"; do run $v; done
