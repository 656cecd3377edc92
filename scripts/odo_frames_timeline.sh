#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p_ot; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ot -- python3 $R/bench.py --workload odometry --steps 60 --warmup 3 --no-cpu-baseline > /tmp/ot.log 2>&1
# (the node with ICET_NODE_TIME_PHASES and the push_many pass come after the timed frames: take frames from the first part of the trace)
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/p_ot/*/*kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(f, newline="")):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    if m: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Queue_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2] == "k_range_count"]
# three consecutive frames of the steady state
lo = idx[30]; hi = idx[33]
t0 = rows[lo][0]
for s, e, n, q in rows[lo:hi]:
    print("%8.1f %7.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n.replace("k_", "")))
PY
