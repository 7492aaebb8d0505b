#!/bin/bash
# round 3 call 2: full GPU suite with the batched SGNS kernel, then its throughput
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3b_tests.log
tail -15 gpurun_out/r3b_tests.log
timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 > gpurun_out/r3b_time_batched_cfg3.log 2>&1 && cat gpurun_out/r3b_time_batched_cfg3.log
