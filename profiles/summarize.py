#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv to the kernels of this
library (plus the rocPRIM sort passes it calls).  usage: summarize.py <kernel_stats.csv> [calls_per_step]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
out = []
for r in rows:
    n = r["Name"]
    if "icet::" in n:
        name = re.search(r"::(k_\w+)", n).group(1)
    elif "radix_sort" in n or "rocprim" in n and "lookback" not in n and "transform" not in n:
        name = "rocprim:" + (re.search(r"wrapped_(\w+?)_config", n).group(1) if re.search(r"wrapped_(\w+?)_config", n) else "other")
    else:
        continue
    out.append((name, int(r["Calls"]), int(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
print("%-34s %7s %12s %10s %10s %10s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
for o in sorted(out, key=lambda t: -t[2]):
    print("%-34s %7d %12.1f %10.1f %10.1f %10.1f" % o)
    tot += o[2]
print("%-34s %7s %12.1f   (%.1f us per step at %g steps)" % ("total", "", tot, tot / steps, steps))
