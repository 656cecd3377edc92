#!/bin/bash
# keyframe launch-shape sweep on the 256-pair batch.  usage through gpurun: kf_sweep.sh "rs_cap=1536" "kf_pts=6" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo -n "$* : "; a=""; for kv in $*; do a="$a --set $kv"; done; python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0.2 $a 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'kf', r['keyframe_ms_per_step'], 'gn', r['gn_loop_ms_per_step'])"; }
run "acc_pts=4"
for v in "$@"; do run $v; done
