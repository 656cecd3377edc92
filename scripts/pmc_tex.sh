#!/bin/bash
# Texture-path counters (TA / TCP / TCC request side) of the default bench's kernels, each group in its own --pmc pass.  Run through gpurun:
#   bash scripts/pmc_tex.sh > gpurun_out/pmc_tex.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-h2d --min-timed-s 0"
i=0; files=""
for grp in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_REQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TCP_GATE_EN2_sum TCP_TCC_WRITE_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1)); rm -rf /tmp/p_tex$i
  echo "pass $i: $grp" >&2; date >&2
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/p_tex$i -- $CMD > /tmp/tex$i.log 2>&1 || { echo "pass $i failed ($grp)"; tail -3 /tmp/tex$i.log; continue; }
  f=$(ls /tmp/p_tex$i/*/*counter_collection.csv | head -1)
  grep -E "Kernel_Name|icet::" $f > /tmp/p_tex$i.csv; files="$files /tmp/p_tex$i.csv"
done
python3 $R/profiles/pmc_summary.py $files
