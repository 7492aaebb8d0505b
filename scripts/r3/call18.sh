#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_sgns_batched_gpu.py -m gpu -q > gpurun_out/r3t_tests.log 2>&1; tail -2 gpurun_out/r3t_tests.log
for h in 0 4096 65536; do HUB_ROWS=$h timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 2>&1 | grep "batched=True"; done | tee gpurun_out/r3t_time_batched_hub_rows_cfg3.log
for h in 4096 65536; do HUB_ROWS=$h timeout -k 10 300 python scripts/r3/hogwild_auc_runs.py 5 batched 0 2>&1 | grep -v "^/opt"; done | tee gpurun_out/r3t_hogwild_auc_batched_hub_rows.log | grep "hub_rows\|mean\|seed=0"
