#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sgns_batched_gpu.py -m gpu -q > gpurun_out/r3h_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3h_tests.log
tail -5 gpurun_out/r3h_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 > gpurun_out/r3h_time_batched_cfg3.log 2>&1 && cat gpurun_out/r3h_time_batched_cfg3.log
timeout -k 10 300 python scripts/r3/time_batched.py cfg3 256 > gpurun_out/r3h_time_batched_cfg3_d256.log 2>&1 && cat gpurun_out/r3h_time_batched_cfg3_d256.log
