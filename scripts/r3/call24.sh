#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_sgns_batched_gpu.py -m gpu -q > gpurun_out/r4a_tests.log 2>&1; tail -2 gpurun_out/r4a_tests.log
timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 2>&1 | grep "batched=True" | tee gpurun_out/r4a_time_batched_10waves.log
timeout -k 10 300 python scripts/r3/time_batched.py cfg3 256 2>&1 | grep "batched=True" | tee -a gpurun_out/r4a_time_batched_10waves.log
