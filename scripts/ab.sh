#!/bin/bash
# Kernel work on the GPU box: bitwise A/B against a baseline build, the GPU parity suite, and bench lines of this build and of
# every icet_amd/lib_exp*/ build.  Usage (through gpurun): bash scripts/ab.sh [notests]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
if [ -f icet_amd/lib_exp_base/libicet_hip.so ]; then timeout -k 10 300 python scripts/cmp_libs.py icet_amd/lib_exp_base/libicet_hip.so 32 2>&1 | tail -3; fi
if [ "$1" != "notests" ]; then timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log; tail -4 gpurun_out/gpu_tests.log | cut -c1-200; fi
bash scripts/exp_libs.sh
