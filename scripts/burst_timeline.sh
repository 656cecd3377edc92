#!/bin/bash
# Device timeline of the LAST frames of `bench.py --workload odometry` (the burst): per launch start offset, duration, stream.  Run through gpurun.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p_bt; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_bt -- python3 $R/bench.py --workload odometry --steps 40 --warmup 3 --no-cpu-baseline > /tmp/bt.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/p_bt/*/*kernel_trace.csv")[0]
rows = []
rd = csv.DictReader(open(f, newline=""))
for r in rd:
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    if m: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
# the last 3 frames: find the last 4 k_range_count launches
idx = [i for i, r in enumerate(rows) if r[2] == "k_range_count"]
lo = idx[-4]
t0 = rows[lo][0]
for s, e, n, q in rows[lo:]:
    print("%8.1f %7.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n.replace("k_", "")))
PY
