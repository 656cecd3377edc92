#!/bin/bash
# the whole -m gpu suite on the GPU box, output kept under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 ${1:-1100} python -m pytest tests -q -m gpu -s -x --durations=15 > gpurun_out/gpu_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gpu_suite.log
tail -40 gpurun_out/gpu_suite.log
