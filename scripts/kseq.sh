#!/bin/bash
# Durations of the LAST step's launches in launch order (which iteration is the slow one?), for the bench arguments given.  Run through gpurun.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p_kq; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_kq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency "$@" > /tmp/kq.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/p_kq/*/*kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(f, newline="")):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    if m: rows.append((int(r["Start_Timestamp"]), m.group(1), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
# the last complete step: from the last k_rs_splitters on
last = max(i for i, r in enumerate(rows) if r[1] == "k_rs_splitters")
print(" ".join("%s:%.0f" % (n.replace("k_", ""), d) for _, n, d in rows[last:]))
PY
