#!/bin/bash
# Bitwise A/B of every icet_amd/lib_exp*/ build against this build, then their bench lines.  Usage (gpurun): bash scripts/ab_all.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for l in icet_amd/lib_exp*/libicet_hip.so; do echo "== $l"; timeout -k 10 300 python scripts/cmp_libs.py $l 16 2>&1 | tail -1; done
bash scripts/exp_libs.sh
