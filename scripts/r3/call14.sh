#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_sgns_batched_gpu.py -m gpu -q > gpurun_out/r3o_tests.log 2>&1
rc=$?; tail -2 gpurun_out/r3o_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 2>&1 | grep "batched=True" | tee gpurun_out/r3o_time_batched_vgprform.log
timeout -k 10 300 python scripts/r3/time_count.py 2>&1 | grep -v "^/opt" | tee gpurun_out/r3o_time_count.log
