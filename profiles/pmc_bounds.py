#!/usr/bin/env python3
"""Per-kernel bound indicators from the rocprofv3 --pmc passes of profiles/collect.sh (one CSV per pass, filtered to this library's
kernels).  For every kernel the WHOLE-BATCH launches are taken (the largest grid: the timed steps run the batch in one part) and the
counters are put in relation to the rows a launch processes:
  wave-instr/row  SQ_INSTS_VALU / rows                  (x 64 = lane instructions per row)
  wait %          SQ_WAIT_ANY / SQ_WAVE_CYCLES           waves parked at s_waitcnt / barrier: latency bound
  stall %         SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES      issue stalls
  active %        SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  LDS conflict    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  rd / wr B/row   2 x FETCH_SIZE, WRITE_SIZE (KB -> B) / rows   (FETCH_SIZE doubled: gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md)
  L2 hit %        TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
usage: pmc_bounds.py rows_per_launch file.csv [file2.csv ...]"""
import csv, re, sys, collections
rows = float(sys.argv[1])
val = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list); grid = collections.defaultdict(list)
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        m = re.search(r"::(k_\w+)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        g = int(r.get("Grid_Size", 0) or 0)
        val[k][r["Counter_Name"]].append((g, float(r["Counter_Value"])))
        dur[k].append((g, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
def big(lst):
    """Mean over the WHOLE-BATCH launches: the largest grid and, inside it, the values within 40 % of the largest (k_gn_accumulate keeps
    its grid when a batch is cut into parts -- its part launches then show as half-size counter values, not as a smaller grid)."""
    gmax = max(g for g, _ in lst); v = [x for g, x in lst if g == gmax]
    top = max(v); w = [x for x in v if x >= 0.6 * top] if top > 0 else v
    return sum(w) / len(w)
print("%-18s %8s %9s %6s %6s %6s %8s %7s %7s %6s" % ("kernel", "us", "winst/row", "wait%", "stall%", "act%", "ldsconf", "rdB/row", "wrB/row", "L2hit%"))
tot = 0.0
for k in sorted(val, key=lambda k: -big(dur[k])):
    c = {n: big(v) for n, v in val[k].items()}
    g = lambda n: c.get(n, 0.0)
    wc = g("SQ_WAVE_CYCLES") or 1.0
    hit = 100 * g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0)
    gmax = max(g_ for g_, _ in dur[k]); dd = sorted(x for g_, x in dur[k] if g_ == gmax); d = dd[len(dd) // 2]; tot += d       # median: one counter pass can slow a kernel 2x
    print("%-18s %8.1f %9.2f %6.1f %6.1f %6.1f %8.2f %7.1f %7.1f %6.1f" % (k, d, g("SQ_INSTS_VALU") / rows, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc,
          100 * g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0), 2048 * g("FETCH_SIZE") / rows, 1024 * g("WRITE_SIZE") / rows, hit))
print("(durations under the profiler; rows per launch = %d)" % rows)
