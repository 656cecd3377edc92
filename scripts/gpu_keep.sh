#!/bin/bash
# round 6: the keep list on the GPU box: its test, then the probe
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "keep_list" -s > gpurun_out/keep_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/keep_tests.log
tail -4 gpurun_out/keep_tests.log
ONLY=2 timeout -k 10 500 python scripts/keep_probe.py "$@" > gpurun_out/keep_probe6.txt 2>&1; cat gpurun_out/keep_probe6.txt
