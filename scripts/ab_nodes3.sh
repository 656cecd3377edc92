#!/bin/bash
# Alternating A/B/C of the sequential callers: this build, icet_amd/lib_exp1 and every other icet_amd/lib_exp_*/ given as arguments (run through gpurun).
cd ${GRAFT_REPO_ROOT:-.}
one() { ICET_HIP_LIB=$2 python bench.py --workload $1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${3:-this}', d['value'], d['ms_per_step'], (d.get('burst') or {}).get('frames_per_s'))"; }
for rep in 1 2 3 4; do one odometry "" this; for l in lib_exp1 "$@"; do one odometry $PWD/icet_amd/$l/libicet_hip.so $l; done; done
