#!/bin/bash
# Long (30-step) bench runs for batch-part settings given as "VAR=VALUE ..." arguments.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo "== $*"; env "$@" python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for v in "$@"; do run $v; done
