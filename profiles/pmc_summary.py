#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counter_collection rows per kernel.  usage: pmc_summary.py file.csv [file2.csv ...]"""
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"::(k_\w+)", n)
        name = m.group(1) if m else ("onesweep" if "onesweep" in n else None)
        if not name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    big = max(dur[k])
    print("%-20s n=%d  dur_us(max)=%.1f" % (k, len(dur[k]), big))
    for c, v in sorted(acc[k].items()):
        print("      %-24s max %.4g   mean %.4g" % (c, max(v), sum(v) / len(v)))
