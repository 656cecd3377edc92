#!/usr/bin/env python3
"""Timeline of the odometry bench under rocprofv3 --kernel-trace: per frame period, busy time per stream / queue, gaps.
usage: odo_timeline.py kernel_trace.csv"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    m = re.search(r"::(k_\w+)", r["Kernel_Name"])
    if not m: continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Queue_Id", r.get("Stream_Id", "?"))))
ev.sort()
# frames: every k_range_count starts one
starts = [i for i, e in enumerate(ev) if e[2] == "k_range_count"]
print("kernels", len(ev), "frames", len(starts))
per = [(ev[starts[i + 1]][0] - ev[starts[i]][0]) / 1e3 for i in range(len(starts) - 1)]
per = per[len(per) // 4:]                                   # steady state
per.sort(); print("frame period us: median %.1f  p10 %.1f  p90 %.1f" % (per[len(per) // 2], per[len(per) // 10], per[9 * len(per) // 10]))
# one steady frame in detail
f = starts[3 * len(starts) // 4]; t0 = ev[f][0]; t1 = ev[starts[3 * len(starts) // 4 + 1]][0]
print("one frame (%.1f us):" % ((t1 - t0) / 1e3))
for s, e, k, q in ev:
    if s >= t0 - 50000 and s < t1: print("  %8.1f .. %8.1f  %6.1f  q=%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, k))
