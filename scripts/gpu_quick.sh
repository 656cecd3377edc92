#!/bin/bash
# Quick GPU check used during kernel work: GPU parity tests, then bench lines for the default env and for each VAR=VALUE argument.
# Usage (through gpurun): bash scripts/gpu_quick.sh [ICET_LDS_SLOTS=312 ...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log; tail -4 gpurun_out/gpu_tests.log | cut -c1-200
run() { echo "== $*"; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'acc_ms', r['avg_launch_ms'], 'frac', r['frac'], 'kf', r['keyframe_ms_per_step'], 'gn', r['gn_loop_ms_per_step'])"; }
run A=1
for v in "$@"; do run $v; done
