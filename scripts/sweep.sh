#!/bin/bash
# Launch-shape sweep of the default bench (timing mode): one line per `--set` combination given as an argument ("a=1,b=2").
# Usage (through gpurun): bash scripts/sweep.sh lds_slots=256 lds_slots=384 acc_blocks=2048,acc_pts=2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo "== $*"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'acc_ms', r['avg_launch_ms'], 'frac', r['frac'], 'kf', r['keyframe_ms_per_step'], 'gn', r['gn_loop_ms_per_step'])"; }
run
for combo in "$@"; do a=(); IFS=',' read -ra kv <<< "$combo"; for x in "${kv[@]}"; do a+=(--set "$x"); done; run "${a[@]}"; done
