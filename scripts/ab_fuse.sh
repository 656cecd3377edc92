#!/bin/bash
# A/B of the shipped library against icet_amd/lib_exp_fuse (built with -DICET_FUSE_DEFAULT=1: small batches run the solve inside the point pass' launch) on the
# sequential callers and the single-pair latency.  Run through gpurun; writes progressively.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for l in "" icet_amd/lib_exp_fuse/libicet_hip.so; do
  if [ -n "$l" ]; then export ICET_HIP_LIB=$PWD/$l; else unset ICET_HIP_LIB; fi
  python bench.py --workload odometry --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=${l:-shipped}', 'odometry fps', d['value'], 'burst', d.get('burst',{}).get('frames_per_s'))"
  python bench.py --workload mapmaker --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=${l:-shipped}', 'mapmaker fps', d['value'])"
done; done
