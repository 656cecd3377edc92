#!/usr/bin/env python3
"""Per-kernel summary from a rocprofv3 `--kernel-trace` CSV (one row per dispatch), split by launch size.
Large device batches are solved in parts on helper streams (icet_capi.hip, batch_parts), while the roofline steps of
bench.py (ICET_FLAG_TIMING) launch every kernel ONCE over the whole batch, alone on the device: the two populations of
k_gn_accumulate are listed separately and the whole-batch one is what bench.py's `roofline.avg_launch_ms` must agree with.
usage: trace_summary.py <kernel_trace.csv> [steps_profiled]"""
import csv, re, sys, collections
path = sys.argv[1]
groups = collections.defaultdict(list)
starts = collections.defaultdict(list)
with open(path, newline="") as f:
    rd = csv.DictReader(f)
    gcol = next(c for c in rd.fieldnames if c.lower().startswith("grid_size") and c.lower().endswith("x"))
    wcol = next(c for c in rd.fieldnames if c.lower().startswith("workgroup_size") and c.lower().endswith("x"))
    for r in rd:
        n = r["Kernel_Name"]
        if "icet::" not in n and "k_range_" not in n and "k_map_" not in n:
            continue
        m = re.search(r"(k_[a-z0-9_]+)", n)
        name = m.group(1) if m else n[:40]
        blocks = int(r[gcol]) // max(int(r[wcol]), 1)
        groups[(name, blocks)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        starts[name].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
by_kernel = collections.defaultdict(list)
for (name, blocks), v in groups.items():
    by_kernel[name].append((blocks, v))
print("%-22s %9s %7s %12s %10s %10s %10s" % ("kernel", "blocks", "calls", "total_us", "avg_us", "min_us", "max_us"))
tot = 0.0
for name in sorted(by_kernel, key=lambda k: -sum(sum(v) for _, v in by_kernel[k])):
    for blocks, v in sorted(by_kernel[name], key=lambda t: -t[0]):
        print("%-22s %9d %7d %12.1f %10.1f %10.1f %10.1f" % (name, blocks, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
        tot += sum(v)
print("total %.1f us over all profiled launches" % tot)
acc, sol = by_kernel.get("k_gn_accumulate"), by_kernel.get("k_gn_solve")
if acc and sol:
    # k_gn_solve runs one block per pair: its largest grid marks the whole-batch launches.  bench.py runs its roofline steps
    # (ICET_FLAG_TIMING: whole batch, one launch per kernel) AFTER the untimed steps, and every accumulate launch is followed by
    # exactly one solve launch, so the LAST n_whole accumulate dispatches are the whole-batch ones (block counts may coincide
    # with those of the part launches, so they cannot be told apart by grid size).
    n_whole = len(max(sol, key=lambda t: t[0])[1])
    v = [d for _, d in sorted(starts["k_gn_accumulate"])[-n_whole:]]
    print("k_gn_accumulate, whole-batch launches (the population bench.py's roofline times): %d calls, avg %.1f us, min %.1f, max %.1f"
          % (len(v), sum(v) / len(v), min(v), max(v)))
    rest = [d for _, d in sorted(starts["k_gn_accumulate"])[:-n_whole]]
    if rest:
        print("k_gn_accumulate, part launches of the untimed steps (run beside other kernels): %d calls, avg %.1f us" % (len(rest), sum(rest) / len(rest)))
