#!/bin/bash
# Per-kernel times of the single-pair (configs[1]) and high-resolution (configs[4]) workloads under rocprofv3 (run through gpurun).
# Outputs: gpurun_out/profiles_small/<tag>_single_pair_kernels.txt, <tag>_highres_kernels.txt
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_one /tmp/p_hi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_one -- python3 $R/bench.py --pairs-per-gpu 1 --steps 50 --warmup 5 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0 > $OUT/one.log 2>&1
grep -E "^\"Name\"|icet::" /tmp/p_one/*/*kernel_stats.csv > $OUT/one_stats.csv
python3 $R/profiles/summarize.py $OUT/one_stats.csv 53 > $OUT/${TAG}_single_pair_kernels.txt
grep "^{\"metric\"" $OUT/one.log | tail -1 >> $OUT/${TAG}_single_pair_kernels.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_hi -- python3 $R/bench.py --workload highres --steps 50 --warmup 5 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0 > $OUT/hi.log 2>&1
grep -E "^\"Name\"|icet::" /tmp/p_hi/*/*kernel_stats.csv > $OUT/hi_stats.csv
python3 $R/profiles/summarize.py $OUT/hi_stats.csv 53 > $OUT/${TAG}_highres_kernels.txt
grep "^{\"metric\"" $OUT/hi.log | tail -1 >> $OUT/${TAG}_highres_kernels.txt
cat $OUT/${TAG}_single_pair_kernels.txt $OUT/${TAG}_highres_kernels.txt
