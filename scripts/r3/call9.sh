#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sgns_gpu.py tests/test_sgns_batched_gpu.py tests/test_partitioned_gpu.py -m gpu -q -s > gpurun_out/r3j_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3j_tests.log
tail -6 gpurun_out/r3j_tests.log; grep "partitioned walking" gpurun_out/r3j_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 > gpurun_out/r3j_time_sgns_cfg3.log 2>&1 && cat gpurun_out/r3j_time_sgns_cfg3.log
