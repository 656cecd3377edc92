#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1; do
for l in "" icet_amd/lib_exp_nofuse/libicet_hip.so; do
  if [ -n "$l" ]; then export ICET_HIP_LIB=$PWD/$l; else unset ICET_HIP_LIB; fi
  python bench.py --workload odometry --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=${l:-fused}', 'odometry fps', d['value'], 'burst', d.get('burst',{}).get('frames_per_s'))"
  python bench.py --workload mapmaker --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=${l:-fused}', 'mapmaker fps', d['value'])"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=${l:-fused}', 'latency', d['latency']['ms_per_pair'], 'loop', d['latency']['gn_loop_ms'], 'highres', d['highres']['ms_per_pair'])"
done; done
