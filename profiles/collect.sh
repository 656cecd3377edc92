#!/bin/bash
# profiles/collect.sh <tag> -- run on the GPU box (through gpurun).  Collects, for the default bench workload:
#   <tag>_kernel_stats.csv / <tag>_kernels.txt   rocprofv3 --kernel-trace --stats   (per-kernel time)
#   <tag>_pmc_*.txt                              SQ / FETCH_SIZE / WRITE_SIZE counters, each in its OWN pass
#   traffic_latest.json                          HBM bytes per k_gn_accumulate launch (FETCH_SIZE doubled as
#                                                MI355X_MICROARCH.md section HBM prescribes for wide coalesced
#                                                reads on gfx950, plus WRITE_SIZE), read by bench.py
# (round 6: the counter passes collect for this library's kernels only -- --kernel-include-regex -- so that the pair generator's thousands of dispatches run unprofiled: a pass takes 20 s instead of 70)
# Outputs go to gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
set -e
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# all 256 pairs distinct: 745 MB of scans, well past the 256 MiB Infinity Cache, so FETCH_SIZE is real HBM traffic
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- $CMD > $OUT/stats.log 2>&1
grep -E "^\"Name\"|icet::" /tmp/p_stats/*/*kernel_stats.csv > $OUT/${TAG}_kernel_stats.csv      # this library's kernels only (torch's generator kernels dropped)
python3 $R/profiles/summarize.py $OUT/${TAG}_kernel_stats.csv 7 > $OUT/${TAG}_kernels.txt
python3 $R/profiles/trace_summary.py $(ls /tmp/p_stats/*/*kernel_trace.csv | head -1) >> $OUT/${TAG}_kernels.txt
grep "^{\"metric\"" $OUT/stats.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "icet" --output-format csv -d /tmp/p_fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --kernel-include-regex "icet" --output-format csv -d /tmp/p_write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "icet" --output-format csv -d /tmp/p_sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "icet" --output-format csv -d /tmp/p_sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "icet" --output-format csv -d /tmp/p_tcc -- $CMD > $OUT/tcc.log 2>&1
for d in p_fetch p_write p_sq1 p_sq2 p_tcc; do
  f=$(ls /tmp/$d/*/*counter_collection.csv | head -1)
  grep -E "Kernel_Name|icet::|onesweep" $f | grep -v "at::native" > /tmp/$d.csv
done
python3 $R/profiles/pmc_summary.py /tmp/p_fetch.csv /tmp/p_write.csv /tmp/p_sq1.csv /tmp/p_sq2.csv /tmp/p_tcc.csv > $OUT/${TAG}_pmc.txt
ROWS=$(python3 -c "import json; d=json.load(open('$OUT/${TAG}_bench_under_rocprof.json')); print(d['config']['points_scan1_mean']*d['config']['pairs_per_gpu'])")
python3 $R/profiles/pmc_bounds.py $ROWS /tmp/p_fetch.csv /tmp/p_write.csv /tmp/p_sq1.csv /tmp/p_sq2.csv /tmp/p_tcc.csv > $OUT/${TAG}_bounds.txt
python3 - <<PY
import csv, json, re, collections
def mean_counter(path, kernel, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    v = [x for x in v if x >= 0.6 * max(v)]   # the whole-batch launches of the timed steps (the parts of the untimed steps are half the size or less)
    return sum(v) / len(v)
fetch_kb = mean_counter("/tmp/p_fetch.csv", "k_gn_accumulate", "FETCH_SIZE")
write_kb = mean_counter("/tmp/p_write.csv", "k_gn_accumulate", "WRITE_SIZE")
traffic = (2.0 * fetch_kb + write_kb) * 1024.0
json.dump({"k_gn_accumulate_bytes_per_launch": traffic, "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
           "note": "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE reports half the bytes of a 16-B/lane coalesced streaming read (MI355X_MICROARCH.md, HBM); separate --pmc passes; workload = default bench (256 distinct pairs)",
           "tag": "$TAG"}, open("$OUT/traffic_latest.json", "w"), indent=1)
print(open("$OUT/traffic_latest.json").read())
PY
cat $OUT/${TAG}_bounds.txt
cat $OUT/${TAG}_kernels.txt
