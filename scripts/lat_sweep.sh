#!/bin/bash
# single-pair / highres latency for launch-shape options.  Run through gpurun.  usage: lat_sweep.sh "name=value name=value" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo -n "$* : "; a=""; for kv in $*; do a="$a --set $kv"; done; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --min-timed-s 0 $a 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('latency', d['latency']['ms_per_pair'], 'kf', d['latency']['keyframe_ms'], 'gn', d['latency']['gn_loop_ms'], '| highres', d['highres']['ms_per_pair'], 'kf', d['highres']['keyframe_ms'], 'gn', d['highres']['gn_loop_ms'])"; }
run "acc_pts=4"
for v in "$@"; do run $v; done
