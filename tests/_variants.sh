#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --distinct 16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'acc_ms', r['avg_launch_ms'], 'frac', r['frac'], 'kf', r['keyframe_ms_per_step'], 'gn', r['gn_loop_ms_per_step'])"; }
for v in "512 4 16" "256 4 16" "256 4 8" "512 4 32" "256 4 32" "1024 4 8"; do
  set -- $v
  rm -f icet_amd/lib/obj/icet_kernels.o
  make -C icet_amd/csrc EXTRA="-DICET_ACC_BLOCK=$1 -DICET_ACC_WAVES=$2 -DICET_ACC_PTS=$3" > /dev/null 2>&1
  echo "== block $1 waves $2 pts $3"
  run
done
