#!/bin/bash
# value (pairs/s, shipped configuration: NOT the timing mode) for batch_parts 1..4 with this build and every icet_amd/lib_exp*/ build.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo -n "$* : "; env $LIBENV python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0.4 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['timed_region']['ms_per_step_min'])"; }
for lib in "" icet_amd/lib_exp*/libicet_hip.so; do
  if [ -n "$lib" ]; then LIBENV="ICET_HIP_LIB=$PWD/$lib"; else LIBENV="A=1"; fi
  echo "== ${lib:-this build}"
  for p in 1 2 3 4; do run --set batch_parts=$p; done
done
