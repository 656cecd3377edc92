#!/bin/bash
# k_gn_solve / k_gn_accumulate durations in launch order for the real batch restricted to one kind of pair (REALKIND=0 frame_804/805, 1 sample_pc_1/2)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for kind in 0 1; do
rm -rf /tmp/p_sq; REAL=1 REALKIND=$kind ONLY=1 PAIRS=${PAIRS:-128} rocprofv3 --kernel-trace --output-format csv -d /tmp/p_sq -- python3 $R/scripts/keep_probe.py > /tmp/sq.log 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("/tmp/p_sq/*/*kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(f, newline="")):
    m = re.search(r"(k_gn_[a-z_]+)", r["Kernel_Name"])
    if m: rows.append((int(r["Start_Timestamp"]), m.group(1), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
print("kind $kind last step:", " ".join("%s:%.0f" % (n.replace("k_gn_", ""), d) for _, n, d in rows[-14:]))
PY
done
