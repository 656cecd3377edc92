#!/bin/bash
# Alternating A/B of the sequential callers between this build and icet_amd/lib_exp1 (run through gpurun): box-to-box and run-to-run noise is
# several per cent, so the two are interleaved.
cd ${GRAFT_REPO_ROOT:-.}
one() { ICET_HIP_LIB=$2 python bench.py --workload $1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${2:-this}', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3 4; do one odometry ""; one odometry $PWD/icet_amd/lib_exp1/libicet_hip.so; done
for rep in 1 2; do one mapmaker ""; one mapmaker $PWD/icet_amd/lib_exp1/libicet_hip.so; done
