#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --distinct 16"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcx -- $CMD > /tmp/pmcx.log 2>&1
f=$(ls /tmp/pmcx/*/*counter_collection.csv | head -1)
grep -E "Kernel_Name|icet::" $f > /tmp/pmcx.csv
python3 $R/profiles/pmc_summary.py /tmp/pmcx.csv | grep -A9 -E "k_rs_bucket_sort|k_scramble_src|k_bin_scatter"
