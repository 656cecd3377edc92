#!/bin/bash
# Times bench.py (timing mode: every kernel alone on the device) against alternative builds of the library
# (icet_amd/lib_exp*/, made with `make -C icet_amd/csrc OUT=../lib_expN EXTRA=-D...`).  Run through gpurun.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo "== $*"; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0.1 $BENCH_ARGS $BENCH_ARGS_ALL 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'acc_ms', r['avg_launch_ms'], 'frac', r['frac'], 'kf', r['keyframe_ms_per_step'], 'gn', r['gn_loop_ms_per_step'])"; }
run A=1
for l in icet_amd/lib_exp*/libicet_hip.so; do
  a=""; [ -f $(dirname $l)/args ] && a=$(cat $(dirname $l)/args)      # per-build bench arguments (e.g. --set lds_slots=240)
  BENCH_ARGS="$BENCH_ARGS_ALL $a" run ICET_HIP_LIB=$PWD/$l
done
